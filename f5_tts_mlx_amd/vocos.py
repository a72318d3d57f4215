"""Vocos mel-24khz vocoder on the HIP engine.

The reference calls the third-party `vocos_mlx.Vocos.decode` (cfm.py:19,399-400,446,471); its source is not part of
the reference tree, so this is a restatement of the upstream gemelo-ai/vocos architecture it ports
(`vocos-mel-24khz`): Conv1d(100->512, k=7) -> LayerNorm -> 8 x ConvNeXt block [dwconv k=7, LayerNorm, Linear 512->1536,
GELU, Linear 1536->512, layer-scale gamma, residual] -> LayerNorm -> Linear(512->1026) -> ISTFT head
(mag = min(exp(.), 1e2), phase -> cos/sin, istft n_fft=1024 hop=256 hann center=True).  PARITY UNPINNED until a real
checkpoint / vocos_mlx output is available; the structure is checked against oracle/vocos_oracle.py + torch.istft.

Every layer is one HIP kernel launch through the C ABI (`f5_op_*`); Python only sequences them, like the reference's
Python-level module tree.
"""
from __future__ import annotations

from typing import Dict

import numpy as np
import torch

from . import engine as E

N_FFT, HOP, N_MELS, DIM, INTER, LAYERS = 1024, 256, 100, 512, 1536, 8


def vocos_param_specs(dim=DIM, inter=INTER, layers=LAYERS, n_mels=N_MELS, n_fft=N_FFT):
    """Upstream (PyTorch) names/shapes of the vocos-mel-24khz backbone + head."""
    specs = [("backbone.embed.weight", (dim, n_mels, 7)), ("backbone.embed.bias", (dim,)),
             ("backbone.norm.weight", (dim,)), ("backbone.norm.bias", (dim,))]
    for i in range(layers):
        p = f"backbone.convnext.{i}."
        specs += [(p + "dwconv.weight", (dim, 1, 7)), (p + "dwconv.bias", (dim,)), (p + "norm.weight", (dim,)),
                  (p + "norm.bias", (dim,)), (p + "pwconv1.weight", (inter, dim)), (p + "pwconv1.bias", (inter,)),
                  (p + "pwconv2.weight", (dim, inter)), (p + "pwconv2.bias", (dim,)), (p + "gamma", (dim,))]
    specs += [("backbone.final_layer_norm.weight", (dim,)), ("backbone.final_layer_norm.bias", (dim,)),
              ("head.out.weight", (n_fft + 2, dim)), ("head.out.bias", (n_fft + 2,))]
    return specs


def synthetic_vocos_weights(seed: int = 7) -> Dict[str, np.ndarray]:
    """Seeded stand-in weights (no checkpoint is reachable offline)."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in vocos_param_specs():
        if name.endswith("norm.weight"):
            w = 1.0 + 0.02 * rng.standard_normal(shape)
        elif name.endswith("gamma"):
            w = 0.125 + 0.01 * rng.standard_normal(shape)
        elif name.endswith("bias"):
            w = 0.02 * rng.standard_normal(shape)
        elif name == "head.out.weight":
            w = rng.standard_normal(shape) * (0.5 / shape[1]) ** 0.5
        else:
            fan_in = int(np.prod(shape[1:]))
            w = rng.standard_normal(shape) * (1.0 / fan_in) ** 0.5
        out[name] = w.astype(np.float32)
    return out


def _split(x: torch.Tensor, two: bool, dtype: torch.dtype = torch.bfloat16):
    """fp32 -> 16-bit MFMA operand (bf16, or fp16 saturated at +-65504 like the device producers) + optional residual."""
    hi = (x.clamp(-65504.0, 65504.0) if dtype == torch.float16 else x).to(dtype)
    lo = (x - hi.to(torch.float32)).to(dtype) if two else None
    return hi.contiguous(), (lo.contiguous() if two else None)


class Vocos:
    def __init__(self, weights: Dict[str, np.ndarray], precision: str = "bf16", device: str | torch.device = "cuda:0"):
        self.lib = E.load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP vocoder needs a GPU device; there is no CPU path")
        self.two = precision == "bf16x3"
        self.nseg = 3 if self.two else 1
        w = {k.replace("backbone.convnext.", "backbone.convnext."): np.asarray(v, dtype=np.float32) for k, v in weights.items()}
        for name, shape in vocos_param_specs():
            if name not in w:
                raise ValueError(f"missing vocoder parameter {name}")
            if name.endswith("embed.weight") and w[name].shape == (DIM, 7, N_MELS):      # MLX conv layout (out, k, in)
                w[name] = np.ascontiguousarray(np.swapaxes(w[name], 1, 2))
            if name.endswith("dwconv.weight") and w[name].shape == (DIM, 7, 1):
                w[name] = np.ascontiguousarray(np.swapaxes(w[name], 1, 2))
            if tuple(w[name].shape) != tuple(shape):
                raise ValueError(f"shape mismatch for {name}: {w[name].shape} vs {shape}")
        dev = self.device
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        # embed conv as a GEMM over im2col rows: W[co][tap][c padded to 128]
        we = np.zeros((DIM, 7, 128), np.float32)
        we[:, :, :N_MELS] = np.transpose(w["backbone.embed.weight"], (0, 2, 1))
        self.w_embed = _split(f(we.reshape(DIM, 7 * 128)), self.two)
        self.b_embed = f(w["backbone.embed.bias"])
        self.norm = (f(w["backbone.norm.weight"]), f(w["backbone.norm.bias"]))
        self.blocks = []
        for i in range(LAYERS):
            p = f"backbone.convnext.{i}."
            self.blocks.append(dict(
                dw_w=f(w[p + "dwconv.weight"].reshape(DIM, 7)), dw_b=f(w[p + "dwconv.bias"]),
                ln_w=f(w[p + "norm.weight"]), ln_b=f(w[p + "norm.bias"]),
                pw1=_split(f(w[p + "pwconv1.weight"]), self.two), b1=f(w[p + "pwconv1.bias"]),
                pw2=_split(f(w[p + "pwconv2.weight"]), self.two), b2=f(w[p + "pwconv2.bias"]), gamma=f(w[p + "gamma"])))
        self.final_norm = (f(w["backbone.final_layer_norm.weight"]), f(w["backbone.final_layer_norm.bias"]))
        wh = np.zeros(((N_FFT + 2 + 127) // 128 * 128, DIM), np.float32)
        wh[: N_FFT + 2] = w["head.out.weight"]
        self.w_head = _split(f(wh), self.two)
        self.b_head = f(w["head.out.bias"])
        self.window = f(np.hanning(N_FFT + 1)[:-1].astype(np.float32))

    @classmethod
    def from_pretrained(cls, path: str, precision: str = "bf16", device: str = "cuda:0") -> "Vocos":
        """Load `model.safetensors` / `pytorch_model.bin`-style weights from a local directory (no network here)."""
        from pathlib import Path
        p = Path(path)
        cands = [p] if p.is_file() else sorted(p.glob("*.safetensors"))
        if not cands:
            raise ValueError(f"Could not find vocoder weights under {path}")
        from safetensors.numpy import load_file
        return cls(load_file(str(cands[0])), precision=precision, device=device)

    def decode(self, mel: torch.Tensor) -> torch.Tensor:
        """mel (b, n, 100) -> wave: 1-D (256*(n-1),) for b == 1 (what generate.py:183 slices), else (b, 256*(n-1))."""
        lib, dev, ns = self.lib, self.device, self.nseg
        mel = mel.to(dev, torch.float32).contiguous()
        B, N, C_ = mel.shape
        assert C_ == N_MELS
        rows = B * N
        P, st = E.ptr, E.stream_ptr(dev)
        bf = lambda *s: torch.empty(s, dtype=torch.bfloat16, device=dev)
        a0, a0l = bf(rows, 7 * 128), (bf(rows, 7 * 128) if self.two else None)
        E.check(lib.f5_op_im2col7(P(mel), P(a0), P(a0l), B, N, N_MELS, st), "im2col7")
        x0 = torch.empty((rows, DIM), device=dev)
        E.check(lib.f5_op_gemm(P(a0), P(a0l), P(self.w_embed[0]), P(self.w_embed[1]), P(self.b_embed), P(x0), P(None), P(None),
                               rows, DIM, 7 * 128, 7 * 128, 7 * 128, DIM, ns, 0, st), "embed gemm")
        x = torch.empty_like(x0)
        E.check(lib.f5_op_layernorm(P(x0), P(self.norm[0]), P(self.norm[1]), P(x), P(None), P(None), rows, DIM, st), "norm")
        h, hl = bf(rows, DIM), (bf(rows, DIM) if self.two else None)
        g, gl = bf(rows, INTER), (bf(rows, INTER) if self.two else None)
        for blk in self.blocks:
            E.check(lib.f5_op_dwconv_ln(P(x), P(blk["dw_w"]), P(blk["dw_b"]), P(blk["ln_w"]), P(blk["ln_b"]), P(h), P(hl), B, N,
                                        DIM, st), "dwconv_ln")
            E.check(lib.f5_op_gemm(P(h), P(hl), P(blk["pw1"][0]), P(blk["pw1"][1]), P(blk["b1"]), P(None), P(g), P(gl), rows,
                                   INTER, DIM, DIM, DIM, INTER, ns, 8, st), "pwconv1")
            E.check(lib.f5_op_gemm_resid_gate(P(g), P(gl), P(blk["pw2"][0]), P(blk["pw2"][1]), P(blk["b2"]), P(blk["gamma"]),
                                              P(None), P(x), rows, DIM, INTER, INTER, INTER, DIM, ns, st), "pwconv2")
        E.check(lib.f5_op_layernorm(P(x), P(self.final_norm[0]), P(self.final_norm[1]), P(None), P(h), P(hl), rows, DIM, st),
                "final norm")
        y = torch.empty((rows, N_FFT + 2), device=dev)
        E.check(lib.f5_op_gemm(P(h), P(hl), P(self.w_head[0]), P(self.w_head[1]), P(self.b_head), P(y), P(None), P(None), rows,
                               N_FFT + 2, DIM, DIM, DIM, N_FFT + 2, ns, 0, st), "head gemm")
        frames = torch.empty((N, N_FFT), device=dev)
        wave = torch.empty((B, HOP * (N - 1)), device=dev)
        for b in range(B):
            E.check(lib.f5_op_istft(P(y[b * N:(b + 1) * N]), N_FFT + 2, P(self.window), P(frames), P(wave[b]), N, N_FFT, HOP, st),
                    "istft")
        return wave[0] if B == 1 else wave

    __call__ = decode
