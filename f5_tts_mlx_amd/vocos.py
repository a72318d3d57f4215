"""Vocos mel-24khz vocoder on the HIP engine.

The reference calls the third-party `vocos_mlx.Vocos.decode` (cfm.py:19,399-400,446,471); its source is not part of
the reference tree, so this is a restatement of the upstream gemelo-ai/vocos architecture it ports
(`vocos-mel-24khz`): Conv1d(100->512, k=7) -> LayerNorm -> 8 x ConvNeXt block [dwconv k=7, LayerNorm, Linear 512->1536,
GELU, Linear 1536->512, layer-scale gamma, residual] -> LayerNorm -> Linear(512->1026) -> ISTFT head
(mag = min(exp(.), 1e2), phase -> cos/sin, istft n_fft=1024 hop=256 hann center=True).  PARITY UNPINNED until a real
checkpoint / vocos_mlx output is available; the structure is checked against oracle/vocos_oracle.py + torch.istft.

The whole decode is ONE C-ABI call (`f5_vocode`, csrc/vocoder.hip): weights arena + workspace owned by this object, the layer
sequence captured as a hipGraph per (batch, frames), the ISTFT of the whole batch in two launches.  This class only owns the
buffers and mirrors `vocos_mlx.Vocos` (`from_pretrained`, `decode`).
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

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


class VocosConfig(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_mels", "dim", "intermediate_dim", "num_layers", "n_fft", "hop_length")]


# state-dict entries of an upstream checkpoint that are buffers of the feature extractor / ISTFT module, not parameters of the
# decode path (the window and the mel filterbank are recomputed)
_IGNORED_PREFIXES = ("feature_extractor.", "head.istft.")


class Vocos:
    def __init__(self, weights: Dict[str, np.ndarray], precision: str = "f16", device: str | torch.device = "cuda:0",
                 use_graph: bool = True):
        if precision not in ("bf16", "bf16x3", "f16"):
            raise ValueError(f"vocoder precision must be bf16, bf16x3 or f16 (got {precision!r})")
        self.lib = E.load_library()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise RuntimeError("the HIP vocoder needs a GPU device; there is no CPU path")
        torch.cuda.set_device(self.device)
        self.precision = precision
        self.use_graph = use_graph
        self._h = C.c_void_p()
        cfg = VocosConfig(N_MELS, DIM, INTER, LAYERS, N_FFT, HOP)
        E.check(self.lib.f5_vocoder_create(C.byref(cfg), E.PRECISIONS[precision], C.byref(self._h)), "f5_vocoder_create")
        nbytes = C.c_size_t()
        E.check(self.lib.f5_vocoder_weights_bytes(self._h, C.byref(nbytes)), "f5_vocoder_weights_bytes")
        self.arena = E._aligned_bytes(nbytes.value, self.device)
        E.check(self.lib.f5_vocoder_set_weights_arena(self._h, E.ptr(self.arena), C.c_size_t(self.arena.numel()),
                                                      E.stream_ptr(self.device)), "f5_vocoder_set_weights_arena")
        want = dict(vocos_param_specs())
        for name, arr in weights.items():
            if name.startswith(_IGNORED_PREFIXES):
                continue
            if name not in want:
                raise ValueError(f"unexpected vocoder parameter {name}")
            a = np.ascontiguousarray(arr, dtype=np.float32)
            shape = (C.c_int64 * a.ndim)(*a.shape)
            E.check(self.lib.f5_vocoder_load_tensor(self._h, name.encode(), a.ctypes.data_as(C.c_void_p), a.ndim, shape),
                    f"f5_vocoder_load_tensor({name})")
        missing = [n for n in want if n not in weights]
        if missing:
            raise ValueError(f"missing vocoder parameter {missing[0]}")
        E.check(self.lib.f5_vocoder_finalize(self._h, E.stream_ptr(self.device)), "f5_vocoder_finalize")
        self._workspace: Optional[torch.Tensor] = None
        self._stream = torch.cuda.Stream(device=self.device)      # hipGraph capture is illegal on the legacy default stream

    def __del__(self):
        try:
            if getattr(self, "_h", None) is not None and self._h.value:
                self.lib.f5_vocoder_destroy(self._h)
                self._h = C.c_void_p()
        except Exception:
            pass

    @classmethod
    def from_pretrained(cls, path: str, precision: str = "f16", device: str = "cuda:0") -> "Vocos":
        """Load `model.safetensors`-style weights (upstream Vocos names) from a local directory (no network here)."""
        from pathlib import Path
        p = Path(path)
        cands = [p] if p.is_file() else sorted(p.glob("*.safetensors"))
        if not cands:
            raise ValueError(f"Could not find vocoder weights under {path}")
        from safetensors.numpy import load_file
        return cls(load_file(str(cands[0])), precision=precision, device=device)

    def _ws(self, B: int, N: int) -> torch.Tensor:
        nbytes = C.c_size_t()
        E.check(self.lib.f5_vocoder_workspace_bytes(self._h, B, N, C.byref(nbytes)), "f5_vocoder_workspace_bytes")
        if self._workspace is None or self._workspace.numel() < nbytes.value:
            self._workspace = None
            self._workspace = E._aligned_bytes(nbytes.value, self.device)
        return self._workspace

    def decode(self, mel: torch.Tensor) -> torch.Tensor:
        """mel (b, n, 100) -> wave: 1-D (256*(n-1),) for b == 1 (what generate.py:183 slices), else (b, 256*(n-1))."""
        mel = mel.to(self.device, torch.float32).contiguous()
        B, N, C_ = mel.shape
        assert C_ == N_MELS
        ws = self._ws(B, N)
        wave = torch.empty((B, HOP * (N - 1)), device=self.device)
        cur = torch.cuda.current_stream(self.device)
        self._stream.wait_stream(cur)
        with torch.cuda.stream(self._stream):
            E.check(self.lib.f5_vocode(self._h, E.ptr(mel), B, N, E.ptr(wave), E.ptr(ws), C.c_size_t(ws.numel()), int(self.use_graph),
                                       C.c_void_p(self._stream.cuda_stream)), "f5_vocode")
        cur.wait_stream(self._stream)
        mel.record_stream(self._stream)
        return wave[0] if B == 1 else wave

    __call__ = decode
