"""CPU ORACLE for the Vocos mel-24khz decoder.  TEST INFRASTRUCTURE ONLY (see oracle/f5_oracle.py header).

PARITY UNPINNED: the reference's vocoder is the third-party package `vocos-mlx` (unpinned, pyproject.toml:42;
call sites cfm.py:19,399-400,446,471) whose source is not in /root/reference and which cannot be installed here.
This file restates the published upstream architecture it ports (gemelo-ai/vocos, `vocos-mel-24khz`):
VocosBackbone (ConvNeXt) + ISTFTHead with `torch.istft(center=True)` semantics — written with stock torch modules
(F.conv1d, F.layer_norm, torch.istft), i.e. independent of the HIP kernels it checks.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F


def decode(weights: Dict[str, np.ndarray], mel: torch.Tensor, dtype=torch.float64, emulate_bf16: bool = False,
           emulate_f16: bool = False) -> torch.Tensor:
    """mel (b, n, 100) -> wave (b, 256 * (n - 1)).  emulate_*: GEMM operands rounded like the engine's bf16 / f16 modes."""
    w = {k: torch.from_numpy(np.asarray(v)).to(dtype) for k, v in weights.items()}
    r = (lambda t: t)
    if emulate_bf16:
        r = lambda t: t.to(torch.bfloat16).to(dtype)
    if emulate_f16:
        r = lambda t: t.clamp(-65504.0, 65504.0).to(torch.float16).to(dtype)
    x = mel.to(dtype).transpose(1, 2)                                                     # (b, 100, n)
    x = F.conv1d(r(x), r(w["backbone.embed.weight"]), w["backbone.embed.bias"], padding=3)
    x = F.layer_norm(x.transpose(1, 2), (x.shape[1],), w["backbone.norm.weight"], w["backbone.norm.bias"], 1e-6)
    n_layers = len([k for k in w if k.endswith("gamma")])
    for i in range(n_layers):
        p = f"backbone.convnext.{i}."
        res = x
        h = F.conv1d(x.transpose(1, 2), w[p + "dwconv.weight"], w[p + "dwconv.bias"], padding=3, groups=x.shape[-1]).transpose(1, 2)
        h = F.layer_norm(h, (h.shape[-1],), w[p + "norm.weight"], w[p + "norm.bias"], 1e-6)
        h = F.gelu(r(h) @ r(w[p + "pwconv1.weight"]).T + w[p + "pwconv1.bias"])
        h = r(h) @ r(w[p + "pwconv2.weight"]).T + w[p + "pwconv2.bias"]
        x = res + w[p + "gamma"] * h
    x = F.layer_norm(x, (x.shape[-1],), w["backbone.final_layer_norm.weight"], w["backbone.final_layer_norm.bias"], 1e-6)
    y = r(x) @ r(w["head.out.weight"]).T + w["head.out.bias"]                             # (b, n, 1026)
    mag, ph = y.transpose(1, 2).chunk(2, dim=1)                                           # (b, 513, n) each
    mag = torch.clip(torch.exp(mag), max=1e2)
    S = mag * (torch.cos(ph) + 1j * torch.sin(ph))
    n_fft = (y.shape[-1] - 2)
    return torch.istft(S, n_fft, n_fft // 4, n_fft, torch.hann_window(n_fft, dtype=dtype), center=True)
