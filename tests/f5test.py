"""Shared helpers for the parity tests (the oracle is imported here and ONLY under tests/)."""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import f5_oracle as O  # noqa: E402
from f5_tts_mlx_amd import engine as E  # noqa: E402
from f5_tts_mlx_amd.weights import TINY, F5TTS_335M, DiTConfig, synthetic_weights  # noqa: E402

DEV = "cuda:0"


# 16-bit MFMA operand type the per-op tests run with: bf16 (default) or fp16 (`with operand_mode("f16")`, which also flips
# the library's f5_op_* entry points to the fp16 build of the kernels)
_OP = {"dtype": torch.bfloat16}


def op_dtype() -> torch.dtype:
    return _OP["dtype"]


class operand_mode:
    def __init__(self, precision: str):
        self.precision = precision

    def __enter__(self):
        self._ctx = E.operand_type(self.precision)
        self._ctx.__enter__()
        _OP["dtype"] = E.operand_dtype(self.precision)
        return self

    def __exit__(self, *exc):
        _OP["dtype"] = torch.bfloat16
        return self._ctx.__exit__(*exc)


def _to_op(x: torch.Tensor) -> torch.Tensor:
    return (x.clamp(-65504.0, 65504.0) if _OP["dtype"] == torch.float16 else x).to(_OP["dtype"])


def split_bf16(x: torch.Tensor):
    """fp32 -> (hi, lo) operand pair (bf16, or fp16 under operand_mode("f16")), same encoding as csrc/op16.hpp f5_split."""
    hi = _to_op(x)
    lo = (x - hi.to(torch.float32)).to(_OP["dtype"])
    return hi.contiguous(), lo.contiguous()


def bf16r(x: torch.Tensor) -> torch.Tensor:
    """round to the operand type and back (bf16 by default)"""
    return _to_op(x).to(torch.float32)


def join(hi: torch.Tensor, lo: torch.Tensor | None) -> torch.Tensor:
    y = hi.to(torch.float32)
    return y if lo is None else y + lo.to(torch.float32)


def stream():
    return E.stream_ptr(torch.device(DEV))


def P(t):
    return E.ptr(t)


def err(a: torch.Tensor, b: torch.Tensor):
    a = a.detach().to("cpu", torch.float64)
    b = b.detach().to("cpu", torch.float64)
    d = (a - b).abs()
    return float(d.max()), float(d.mean()), float(b.abs().mean())


def report(name, a, b):
    mx, mean, ref = err(a, b)
    print(f"[parity] {name}: max_abs={mx:.3e} mean_abs={mean:.3e} ref_mean_abs={ref:.3e}")
    return mx, mean, ref


def rng(seed=0):
    return np.random.default_rng(seed)


def randn(r, *shape, scale=1.0):
    return torch.from_numpy((r.standard_normal(shape) * scale).astype(np.float32))


def synth_inputs(cfg: DiTConfig, B: int, N: int, nt: int, n_ref: int, seed=0, ragged=False):
    """Seeded synthetic sample() inputs in the shape of SURVEY §8(d)."""
    r = rng(seed)
    cond = randn(r, B, n_ref, cfg.mel_dim)
    text = torch.from_numpy(r.integers(0, cfg.text_num_embeds, (B, nt)).astype(np.int32))
    durations = [N] * B
    if ragged and B > 1:
        durations = [N - 7 * i for i in range(B)]
        for i in range(1, B):
            text[i, nt - 3 * i:] = -1
    y0 = np.zeros((B, N, cfg.mel_dim), np.float32)
    for i in range(B):
        z = r.standard_normal((cfg.mel_dim, durations[i])).astype(np.float32)
        y0[i, :durations[i]] = z.T
    return cond, text, durations, torch.from_numpy(y0)
