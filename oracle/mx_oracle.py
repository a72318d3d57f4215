"""CPU restatement of the MX (OCP microscaling) fp8 quantisation used by the engine's `mxfp8` precision mode.

TEST INFRASTRUCTURE ONLY (imported by tests/ and bench.py's checker paths, never by the product).  There is no reference
counterpart: the reference's reduced-precision mode is MLX int4/int8 weight quantisation (cfm.py:510-515); BASELINE.json
configs[4] asks for CDNA4 fp8 MFMA instead.  Element format OCP e4m3fn (max 448), one power-of-two scale (E8M0) per 32
consecutive elements of the contraction axis: e = ceil(log2(amax / 448)), so amax / 2^e lies in (224, 448] and nothing
saturates.  `amax / 448` is evaluated as one fp32 multiply by the fp32-rounded reciprocal, exactly as the kernels do.
"""
from __future__ import annotations

import numpy as np
import torch

RCP448 = np.float32(float.fromhex("0x1.24924ap-9"))
BLOCK = 32


def mx_scale_bytes(x: torch.Tensor) -> torch.Tensor:
    """E8M0 byte per 32-block of the last axis (uint8), x fp32 with last dim % 32 == 0."""
    xb = x.to(torch.float32).reshape(*x.shape[:-1], x.shape[-1] // BLOCK, BLOCK)
    amax = xb.abs().amax(dim=-1)
    y = (amax * torch.tensor(RCP448)).contiguous()
    bits = y.view(torch.int32)
    exp = (bits >> 23) & 255
    e = exp + ((bits & 0x7FFFFF) != 0).to(torch.int32)
    e = torch.where(exp == 0, torch.zeros_like(e), e)
    return torch.clamp(e, max=254).to(torch.uint8)


def mx_quantize(x: torch.Tensor):
    """-> (q uint8 view of e4m3 bytes, same shape as x; scales uint8 [..., K/32])."""
    e8 = mx_scale_bytes(x)
    inv = torch.pow(torch.tensor(2.0, dtype=torch.float64), (127 - e8.to(torch.int32)).to(torch.float64)).to(torch.float32)
    inv = torch.where(e8 == 254, torch.zeros_like(inv), inv)        # the kernel's 2^-127 is a flushed denormal
    xb = x.to(torch.float32).reshape(*x.shape[:-1], x.shape[-1] // BLOCK, BLOCK) * inv[..., None]
    q = torch.clamp(xb, -448.0, 448.0).to(torch.float8_e4m3fn).view(torch.uint8).reshape(x.shape)
    return q, e8


def mx_dequantize(q: torch.Tensor, e8: torch.Tensor, dtype=torch.float64) -> torch.Tensor:
    v = q.view(torch.float8_e4m3fn).to(dtype).reshape(*q.shape[:-1], q.shape[-1] // BLOCK, BLOCK)
    s = torch.pow(torch.tensor(2.0, dtype=dtype), (e8.to(torch.int32) - 127).to(dtype))
    return (v * s[..., None]).reshape(q.shape)


def mx_round(x: torch.Tensor) -> torch.Tensor:
    """quantise-dequantise (what an MX-fp8 operand holds), in x's dtype"""
    q, e8 = mx_quantize(x)
    return mx_dequantize(q, e8, torch.float64).to(x.dtype)
