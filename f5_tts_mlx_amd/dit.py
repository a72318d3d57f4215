"""DiT backbone handle (reference: f5_tts_mlx/dit.py:331-401).

The reference `DiT` is an nn.Module tree evaluated op by op; here it is a configuration + weight set
bound to one HIP `Engine`.  `DiT.__call__` keeps the reference signature for single forwards (tests,
diagnostics); `F5TTS.sample` drives the whole ODE solve through `Engine.sample` instead of calling
the model 2 x NFE times from Python.
"""
from __future__ import annotations

from typing import Dict, Optional

import numpy as np
import torch

from .engine import Engine
from .weights import DiTConfig


class DiT:
    def __init__(self, *, dim, depth=8, heads=8, dim_head=64, dropout=0.0, ff_mult=4, mel_dim=100, text_num_embeds=256,
                 text_dim=None, text_mask_padding=True, conv_layers=0, precision: str = "f16",
                 device: str | torch.device = "cuda:0"):
        if text_dim is None:
            text_dim = mel_dim
        if dropout != 0.0:
            raise NotImplementedError("dropout is training-only")
        if conv_layers < 0:
            raise ValueError("conv_layers must be >= 0")
        self.cfg = DiTConfig(dim=dim, depth=depth, heads=heads, dim_head=dim_head, ff_mult=ff_mult, mel_dim=mel_dim,
                             text_num_embeds=text_num_embeds, text_dim=text_dim, conv_layers=conv_layers,
                             conv_pos_groups=dim // 64, text_mask_padding=bool(text_mask_padding))
        self.dim = dim
        self.depth = depth
        self.precision = precision
        self.device = torch.device(device)
        self._engine: Optional[Engine] = None

    @classmethod
    def from_config(cls, cfg: DiTConfig, precision="f16", device="cuda:0") -> "DiT":
        m = cls(dim=cfg.dim, depth=cfg.depth, heads=cfg.heads, dim_head=cfg.dim_head, ff_mult=cfg.ff_mult, mel_dim=cfg.mel_dim,
                text_num_embeds=cfg.text_num_embeds, text_dim=cfg.text_dim, conv_layers=cfg.conv_layers,
                text_mask_padding=getattr(cfg, "text_mask_padding", True), precision=precision, device=device)
        m.cfg = cfg
        return m

    @property
    def engine(self) -> Engine:
        if self._engine is None:
            self._engine = Engine(self.cfg, precision=self.precision, device=self.device)
        return self._engine

    def load_weights(self, weights: Dict[str, np.ndarray]) -> None:
        """weights: reference (MLX-layout) names -> fp32 arrays (weights.param_specs)."""
        self.engine.load_weights({k: np.asarray(v) for k, v in weights.items()})

    def __call__(self, x, cond, text, time, drop_audio_cond, drop_text, mask=None) -> torch.Tensor:
        """dit.py:374-401 — one forward, any (drop_audio_cond, drop_text).  The engine evaluates two branches per call: the
        conditional one (False, False) and a second one that drops the text and, by default, the audio conditioning (True, True).
        (True, False) — audio dropped, text kept (training, cfm.py:222-225) — is the conditional branch on a zeroed `cond`
        (dit.py:245-247); (False, True) — text dropped, audio kept; the reference never produces it (drop_text forces
        drop_audio_cond, cfm.py:225) but its `DiT` accepts it — is the second branch with the engine option `null_keeps_cond`.
        `time` may be a scalar or a (b,) tensor; rows with different times run as separate batch-1 forwards at the SAME padded
        length with their own row of `mask`, which is exact (no cross-utterance arithmetic in the DiT; GRN / conv-pos see the
        same padding)."""
        drop_audio_cond, drop_text = _flag(drop_audio_cond), _flag(drop_text)
        if drop_text and not drop_audio_cond:
            self.engine.set_option("null_keeps_cond", 1)
            try:
                return self._forward(x, cond, text, time, False, True, mask)
            finally:
                self.engine.set_option("null_keeps_cond", 0)
        return self._forward(x, cond, text, time, drop_audio_cond, drop_text, mask)

    def _forward(self, x, cond, text, time, drop_audio_cond, drop_text, mask=None) -> torch.Tensor:
        B, N, _ = x.shape
        x = x.to(self.device, torch.float32).contiguous()
        cond = cond.to(self.device, torch.float32).contiguous()
        text = text.to(self.device, torch.int32).contiguous()
        if drop_audio_cond and not drop_text:
            cond = torch.zeros_like(cond)
        if torch.is_tensor(time) and time.numel() > 1:
            times = [float(v) for v in time.reshape(-1).tolist()]
            if len(times) != B:
                raise ValueError(f"time has {len(times)} entries for a batch of {B}")
        else:
            times = [float(time if not torch.is_tensor(time) else time.reshape(-1)[0])] * B
        if len(set(times)) > 1:
            outs = [self._forward(x[i:i + 1], cond[i:i + 1], text[i:i + 1], times[i], drop_audio_cond, drop_text,
                                  None if mask is None else mask[i:i + 1]) for i in range(B)]
            return torch.cat(outs, dim=0)
        if mask is not None:
            durations = mask.sum(dim=-1).to(torch.int32).tolist()
        else:
            durations = [N] * B
        # `cond` arrives already masked (step_cond), so the conditioning length is the full sequence
        pred, null = self.engine.dit_forward(x, text, cond, [N] * B, durations, times[0],
                                             cfg_strength=2.0 if drop_text else 0.0, use_mask=mask is not None)
        return null if drop_text else pred


def _flag(v) -> bool:
    return bool(v.reshape(-1)[0]) if torch.is_tensor(v) else bool(v)
