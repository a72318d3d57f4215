"""CPU ORACLE for the duration predictor (reference: f5_tts_mlx/duration.py:44-260).  TEST INFRASTRUCTURE ONLY.
Pinned against the reference's duration.py executed over oracle/mlx_shim.py (tests/golden/ref_duration.npz); MLX's own
arithmetic stays unpinned (see oracle/f5_oracle.py).  Reuses the DiT oracle's primitives for TextEmbedding / ConvNeXt / attention."""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import numpy as np
import torch
import torch.nn.functional as F

from . import f5_oracle as O


def predict(weights: Dict[str, np.ndarray], inp: torch.Tensor, text: torch.Tensor, lens=None, dim=512, depth=8, heads=8,
            text_dim=512, conv_layers=2, dtype=torch.float64, emulate_bf16=False) -> torch.Tensor:
    """DurationPredictor.__call__ (duration.py:192-251) with return_loss=False; inp (b, n, mel) -> seconds (b,)."""
    cfg = SimpleNamespace(dim=dim, depth=depth, heads=heads, dim_head=64, mel_dim=inp.shape[-1], text_dim=text_dim,
                          conv_layers=conv_layers, conv_pos_kernel=31, conv_pos_groups=16, freq_embed_dim=256, text_max_pos=4096)
    orc = O.DiTOracle(cfg, weights, dtype=dtype, emulate_bf16=emulate_bf16)
    inp = inp.to(dtype)
    batch, seq_len = inp.shape[:2]
    if seq_len < text.shape[1]:                                           # duration.py:218-220
        inp = F.pad(inp, (0, 0, 0, text.shape[1] - seq_len))
        seq_len = text.shape[1]
    if lens is None:
        lens = torch.full((batch,), seq_len)
    mask = O.lens_to_mask(torch.as_tensor(lens), seq_len)
    inp = torch.where(mask[..., None], inp, torch.zeros_like(inp))       # :243-245
    # TextEmbedding with mask_padding=False (dit.py:196-229: no text_mask, plain text_blocks)
    t = text.to(torch.int64) + 1
    t = F.pad(t[:, :seq_len], (0, seq_len - text.shape[1]), value=0)
    te = orc.w["transformer.text_embed.text_embed.weight"][t] + orc.freqs_cis[:seq_len][None]
    for i in range(conv_layers):
        te = orc.convnext_block(te, i)
    x = orc.linear(torch.cat((inp, te), dim=-1), "transformer.input_embed.proj")             # duration.py:56
    x = orc.conv_pos_embed(x) + x                                                              # :57
    rope = O.rotary_freqs(64, seq_len)
    for i in range(depth):                                                                     # DurationBlock :81-94
        p = f"transformer.transformer_blocks.{i}."
        x = x + orc.attention(orc.layer_norm(x), i, None, rope)
        h = O._gelu_tanh(orc.linear(orc.layer_norm(x), p + "ff.ff.layers.0.layers.0"))
        x = x + orc.linear(h, p + "ff.ff.layers.2")
    g = orc.w["transformer.norm_out.weight"]                                                   # nn.RMSNorm, eps 1e-5
    x = x * torch.rsqrt((x * x).mean(dim=-1, keepdim=True) + 1e-5) * g
    m = mask[..., None].to(dtype)
    mean = (x * m).sum(dim=1) / torch.clamp(m.sum(dim=1), min=1)                               # utils.py:82-90
    pred = mean @ orc.w["to_pred.layers.0.weight"].T                                           # (b, 1)
    return F.softplus(pred)[:, 0]
