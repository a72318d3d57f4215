"""Duration predictor (reference: f5_tts_mlx/duration.py) on the HIP engine — SURVEY.md §8(f) "next #1".

`generate()` without `--duration` reaches it through `F5TTS.predict_duration` (cfm.py:253-262, :307-308).  The model is the
reference's `DurationPredictor(DurationTransformer(dim=512, depth=8, heads=8, text_dim=512, ff_mult=2, conv_layers=2))`
(cfm.py:429-438): TextEmbedding WITHOUT padding mask, `Linear(mel+text -> dim)` + ConvPositionEmbedding, 8 pre-LN blocks
(plain LayerNorm, no adaLN, no gates), RMSNorm, masked mean over time, `Linear(dim -> 1, no bias)` + Softplus = seconds.
Every layer is one HIP kernel launch through the C ABI (`f5_op_*`); Python only sequences them.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional

import numpy as np
import torch

from . import engine as E
from .audio import MelSpec
from .utils import default, exists, lens_to_mask, list_str_to_idx, list_str_to_tensor

SAMPLE_RATE = 24_000
HOP_LENGTH = 256
SAMPLES_PER_SECOND = SAMPLE_RATE / HOP_LENGTH


def duration_param_specs(dim=512, depth=8, mel_dim=100, text_num_embeds=2545, text_dim=512, conv_layers=2, ff_mult=2):
    """Reference (MLX-layout) parameter names/shapes of DurationPredictor (duration.py:97-190)."""
    TF, FF = text_dim * 2, dim * ff_mult
    specs = [("transformer.text_embed.text_embed.weight", (text_num_embeds + 1, text_dim))]
    for i in range(conv_layers):
        q = f"transformer.text_embed.text_blocks.layers.{i}."
        specs += [(q + "dwconv.weight", (text_dim, 7, 1)), (q + "dwconv.bias", (text_dim,)), (q + "norm.weight", (text_dim,)),
                  (q + "norm.bias", (text_dim,)), (q + "pwconv1.weight", (TF, text_dim)), (q + "pwconv1.bias", (TF,)),
                  (q + "grn.gamma", (1, 1, TF)), (q + "grn.beta", (1, 1, TF)), (q + "pwconv2.weight", (text_dim, TF)),
                  (q + "pwconv2.bias", (text_dim,))]
    specs += [("transformer.input_embed.proj.weight", (dim, mel_dim + text_dim)), ("transformer.input_embed.proj.bias", (dim,))]
    for j in (0, 2):
        specs += [(f"transformer.input_embed.conv_pos_embed.conv1d.layers.{j}.weight", (dim, 31, dim // 16)),
                  (f"transformer.input_embed.conv_pos_embed.conv1d.layers.{j}.bias", (dim,))]
    for i in range(depth):
        q = f"transformer.transformer_blocks.{i}."
        for nm in ("to_q", "to_k", "to_v"):
            specs += [(q + f"attn.{nm}.weight", (dim, dim)), (q + f"attn.{nm}.bias", (dim,))]
        specs += [(q + "attn.to_out.layers.0.weight", (dim, dim)), (q + "attn.to_out.layers.0.bias", (dim,)),
                  (q + "ff.ff.layers.0.layers.0.weight", (FF, dim)), (q + "ff.ff.layers.0.layers.0.bias", (FF,)),
                  (q + "ff.ff.layers.2.weight", (dim, FF)), (q + "ff.ff.layers.2.bias", (dim,))]
    specs += [("transformer.norm_out.weight", (dim,)), ("to_pred.layers.0.weight", (1, dim))]
    return specs


def synthetic_duration_weights(seed: int = 11, **kw) -> Dict[str, np.ndarray]:
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in duration_param_specs(**kw):
        if name.endswith("norm.weight") or name.endswith("norm_out.weight"):
            w = 1.0 + 0.02 * rng.standard_normal(shape)
        elif name.endswith("bias"):
            w = 0.02 * rng.standard_normal(shape)
        elif "grn." in name:
            w = 0.1 * rng.standard_normal(shape)
        elif name.endswith("text_embed.weight"):
            w = rng.standard_normal(shape)
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


class DurationTransformer:
    """duration.py:97-158 — configuration + device weights."""

    def __init__(self, *, dim, depth=8, heads=8, dim_head=64, dropout=0.0, ff_mult=4, mel_dim=100, text_num_embeds=256,
                 text_dim=None, conv_layers=0, precision: str = "bf16", device: str | torch.device = "cuda:0"):
        if text_dim is None:
            text_dim = mel_dim
        assert heads * dim_head == dim and dim_head == 64 and dim % 256 == 0 and text_dim % 256 == 0 and conv_layers > 0
        assert (dim // 16) in (32, 64), "conv position embedding groups of 32 or 64 channels"
        self.dim, self.depth, self.heads, self.ff_dim = dim, depth, heads, int(dim * ff_mult)
        self.mel_dim, self.text_num_embeds, self.text_dim, self.conv_layers = mel_dim, text_num_embeds, text_dim, conv_layers
        self.device = torch.device(device)
        self.precision = precision
        self.op_dtype = E.operand_dtype(precision)        # bf16, or fp16 for precision "f16"
        self.two = precision == "bf16x3"
        self.nseg = 3 if self.two else 1
        self.w = None

    def load_weights(self, weights: Dict[str, np.ndarray]) -> None:
        dev, two = self.device, self.two
        _split_op = lambda x, two_: _split(x, two_, self.op_dtype)
        f = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).to(dev)
        pad128 = lambda m: np.concatenate([m, np.zeros(((-m.shape[0]) % 128, m.shape[1]), np.float32)], axis=0)
        g = {k.replace("duration_predictor.", ""): np.asarray(v, dtype=np.float32) for k, v in weights.items()}
        for name, shape in duration_param_specs(self.dim, self.depth, self.mel_dim, self.text_num_embeds, self.text_dim,
                                                self.conv_layers, self.ff_dim // self.dim):
            if name not in g or tuple(g[name].shape) != tuple(shape):
                raise ValueError(f"duration predictor: missing / mis-shaped parameter {name}")
        W = {}
        W["table"] = f(g["transformer.text_embed.text_embed.weight"])
        W["pos"] = torch.empty((4096, self.text_dim), device=dev)
        E.check(E.load_library().f5_op_text_pos_table(E.ptr(W["pos"]), 4096, self.text_dim, E.stream_ptr(dev)))
        W["tblocks"] = []
        for i in range(self.conv_layers):
            q = f"transformer.text_embed.text_blocks.layers.{i}."
            W["tblocks"].append(dict(
                dw_w=f(g[q + "dwconv.weight"].reshape(self.text_dim, 7)), dw_b=f(g[q + "dwconv.bias"]), ln_w=f(g[q + "norm.weight"]),
                ln_b=f(g[q + "norm.bias"]), pw1=_split_op(f(pad128(g[q + "pwconv1.weight"])), two), b1=f(g[q + "pwconv1.bias"]),
                gamma=f(g[q + "grn.gamma"].reshape(-1)), beta=f(g[q + "grn.beta"].reshape(-1)),
                pw2=_split_op(f(pad128(g[q + "pwconv2.weight"])), two), b2=f(g[q + "pwconv2.bias"])))
        wp = g["transformer.input_embed.proj.weight"]                      # (dim, mel + text): input order x | text
        wcat = np.zeros((self.dim, 128 + self.text_dim), np.float32)
        wcat[:, : self.mel_dim] = wp[:, : self.mel_dim]
        wcat[:, 128:] = wp[:, self.mel_dim:]
        W["proj"] = _split_op(f(pad128(wcat)), two)
        W["bproj"] = f(g["transformer.input_embed.proj.bias"])
        cg = self.dim // 16
        W["conv"] = []
        for j in (0, 2):
            w = g[f"transformer.input_embed.conv_pos_embed.conv1d.layers.{j}.weight"]      # (dim, 31, cg)
            if cg == 64:
                wbd = w
            else:   # 32-channel groups: pair them into 64-channel super groups with block-diagonal weights
                wbd = np.zeros((self.dim, 31, 64), np.float32)
                for o in range(self.dim):
                    off = ((o // 32) % 2) * 32
                    wbd[o, :, off:off + 32] = w[o]
            W["conv"].append((_split_op(f(wbd.reshape(self.dim, 31 * 64)), two),
                              f(g[f"transformer.input_embed.conv_pos_embed.conv1d.layers.{j}.bias"])))
        W["blocks"] = []
        for i in range(self.depth):
            q = f"transformer.transformer_blocks.{i}."
            wqkv = np.concatenate([g[q + f"attn.{n}.weight"] for n in ("to_q", "to_k", "to_v")], axis=0)
            bqkv = np.concatenate([g[q + f"attn.{n}.bias"] for n in ("to_q", "to_k", "to_v")], axis=0)
            W["blocks"].append(dict(
                qkv=_split_op(f(pad128(wqkv)), two), bqkv=f(bqkv), o=_split_op(f(pad128(g[q + "attn.to_out.layers.0.weight"])), two),
                bo=f(g[q + "attn.to_out.layers.0.bias"]), ff1=_split_op(f(pad128(g[q + "ff.ff.layers.0.layers.0.weight"])), two),
                bff1=f(g[q + "ff.ff.layers.0.layers.0.bias"]), ff2=_split_op(f(pad128(g[q + "ff.ff.layers.2.weight"])), two),
                bff2=f(g[q + "ff.ff.layers.2.bias"])))
        W["norm_out"] = f(g["transformer.norm_out.weight"])
        W["to_pred"] = f(g["to_pred.layers.0.weight"].reshape(-1))
        W["zeros"] = torch.zeros(self.dim, device=dev)
        W["ones"] = torch.ones(self.dim, device=dev)
        self.w = W


class DurationPredictor:
    """duration.py:161-260 (inference path; `return_loss=True` is training and out of scope)."""

    def __init__(self, transformer: DurationTransformer, num_channels=None, mel_spec_kwargs: dict = dict(),
                 vocab_char_map: dict[str, int] | None = None):
        self._mel_spec = MelSpec(**dict(dict(device=getattr(transformer, "device", None)), **mel_spec_kwargs))
        self.num_channels = default(num_channels, self._mel_spec.n_mels)
        self.transformer = transformer
        self.dim = transformer.dim
        self._vocab_char_map = vocab_char_map

    def load_weights(self, weights) -> None:
        self.transformer.load_weights(dict(weights))

    def __call__(self, inp: torch.Tensor, text, *, lens: Optional[torch.Tensor] = None, return_loss=False) -> torch.Tensor:
        if return_loss:
            raise NotImplementedError("training loss (duration.py:229-260) is out of scope")
        T = self.transformer
        if T.w is None:
            raise RuntimeError("duration predictor weights not loaded")
        lib, dev, ns, W = E.load_library(), T.device, T.nseg, T.w
        inp = torch.as_tensor(inp)
        if inp.ndim == 2:                                   # raw wave (duration.py:203-206)
            inp = self._mel_spec(inp)
            assert inp.shape[-1] == self.num_channels
        inp = inp.to(dev, torch.float32)
        batch, seq_len = inp.shape[:2]
        if isinstance(text, list):
            text = list_str_to_idx(text, self._vocab_char_map) if exists(self._vocab_char_map) else list_str_to_tensor(text)
            assert text.shape[0] == batch
        text = torch.as_tensor(text).to(torch.int32)
        if seq_len < text.shape[1]:                         # duration.py:218-220
            pad = torch.zeros((batch, text.shape[1] - seq_len, inp.shape[2]), device=dev)
            inp = torch.cat([inp, pad], dim=1)              # plumbing: zero padding of the input buffer
            seq_len = text.shape[1]
        if not exists(lens):
            lens = torch.full((batch,), seq_len, dtype=torch.int64)
        mask = lens_to_mask(torch.as_tensor(lens).to("cpu", torch.int64), length=seq_len)          # (b, n) bool, host
        mask_d = mask.to(torch.uint8).contiguous().to(dev)
        inp = inp.contiguous()
        text_d = text.contiguous().to(dev)

        with E.operand_type(T.precision):                  # the f5_op_* entry points take this model's operand type
            return self._run_ops(lib, dev, ns, W, T, inp, text, text_d, mask_d, batch, seq_len)

    def _run_ops(self, lib, dev, ns, W, T, inp, text, text_d, mask_d, batch, seq_len):
        B, N, D, Dt, H = batch, seq_len, T.dim, T.text_dim, T.heads
        rows, npad = B * N, (N + 63) // 64 * 64
        P, st = E.ptr, E.stream_ptr(dev)
        two = T.two
        bf = lambda *s: torch.empty(s, dtype=T.op_dtype, device=dev)
        bfz = lambda *s: torch.zeros(s, dtype=T.op_dtype, device=dev)
        lo = lambda t: t if two else None
        ck = E.check

        # ---- text path, no padding mask (duration.py:116-118, dit.py:226-227)
        te = torch.empty((2, B, N, Dt), device=dev)
        ids = torch.empty((2, B, N), dtype=torch.int32, device=dev)
        keep = torch.empty((2, B, N), dtype=torch.uint8, device=dev)
        ck(lib.f5_op_text_embed_nomask(P(text_d), text.shape[1], P(W["table"]), P(W["pos"]), 4096, P(te), P(ids), P(keep), B, N, Dt, st))
        t_cur, t_nxt = te[0].reshape(rows, Dt), torch.empty((rows, Dt), device=dev)
        tln, tlnl = bf(rows, Dt), lo(bf(rows, Dt))
        tg = torch.empty((rows, 2 * Dt), device=dev)
        tg2, tg2l = bf(rows, 2 * Dt), lo(bf(rows, 2 * Dt))
        scratch = torch.empty(lib.f5_op_grn_scratch_floats(B, N, 2 * Dt), device=dev)
        for blk in W["tblocks"]:
            ck(lib.f5_op_dwconv_ln(P(t_cur), P(blk["dw_w"]), P(blk["dw_b"]), P(blk["ln_w"]), P(blk["ln_b"]), P(tln), P(tlnl), B, N, Dt, st))
            ck(lib.f5_op_gemm(P(tln), P(tlnl), P(blk["pw1"][0]), P(blk["pw1"][1]), P(blk["b1"]), P(tg), P(None), P(None), rows,
                              2 * Dt, Dt, Dt, Dt, 2 * Dt, ns, 3, st))
            ck(lib.f5_op_grn(P(tg), P(blk["gamma"]), P(blk["beta"]), P(scratch), P(tg2), P(tg2l), B, N, 2 * Dt, st))
            ck(lib.f5_op_gemm_resid_keep(P(tg2), P(tg2l), P(blk["pw2"][0]), P(blk["pw2"][1]), P(blk["b2"]), P(t_cur), P(None),
                                         P(t_nxt), rows, Dt, 2 * Dt, 2 * Dt, 2 * Dt, Dt, ns, st))
            t_cur, t_nxt = t_nxt, t_cur

        # ---- input embedding: proj(concat(masked mel, text)) + conv_pos_embed (duration.py:44-58, :243-247)
        K0 = 128 + Dt
        a0, a0l = bfz(rows, K0), lo(bfz(rows, K0))
        ck(lib.f5_op_pack_bf16(P(inp), P(mask_d), P(a0), P(a0l), rows, self.num_channels, K0, 0, st))
        ck(lib.f5_op_pack_bf16(P(t_cur), P(None), P(a0), P(a0l), rows, Dt, K0, 128, st))
        x = torch.empty((rows, D), device=dev)
        ck(lib.f5_op_gemm(P(a0), P(a0l), P(W["proj"][0]), P(W["proj"][1]), P(W["bproj"]), P(x), P(None), P(None), rows, D, K0, K0, K0,
                          D, ns, 0, st))
        xb, xbl, c1, c1l = bf(rows, D), lo(bf(rows, D)), bf(rows, D), lo(bf(rows, D))
        ck(lib.f5_op_pack_bf16(P(x), P(None), P(xb), P(xbl), rows, D, D, 0, st))
        (cw0, cb0), (cw1, cb1) = W["conv"]
        ck(lib.f5_op_convpos(P(xb), P(xbl), P(cw0[0]), P(cw0[1]), P(cb0), P(c1), P(c1l), P(None), B, N, D, D // 64, 31, ns, 0, st))
        ck(lib.f5_op_convpos(P(c1), P(c1l), P(cw1[0]), P(cw1[1]), P(cb1), P(None), P(None), P(x), B, N, D, D // 64, 31, ns, 1, st))

        # ---- pre-LN transformer blocks without modulation / gates / masks (duration.py:64-94)
        cos_t, sin_t = torch.empty((N, 32), device=dev), torch.empty((N, 32), device=dev)
        ck(lib.f5_op_rope_table(P(cos_t), P(sin_t), N, 64, st))
        h, hl = bf(rows, D), lo(bf(rows, D))
        qk, qkl = bf(rows, 2 * D), lo(bf(rows, 2 * D))
        vt, vtl = bfz(B * H, 64, npad), lo(bfz(B * H, 64, npad))
        ao, aol = bf(rows, D), lo(bf(rows, D))
        ffh, ffhl = bf(rows, T.ff_dim), lo(bf(rows, T.ff_dim))
        for blk in W["blocks"]:
            ck(lib.f5_op_ln_modulate(P(x), P(W["zeros"]), P(W["zeros"]), P(h), P(hl), rows, D, st))
            ck(lib.f5_op_qkv_rope(P(h), P(hl), P(blk["qkv"][0]), P(blk["qkv"][1]), P(blk["bqkv"]), P(cos_t), P(sin_t), P(qk), P(qkl),
                                  P(vt), P(vtl), B, N, npad, H, D, ns, st))
            ck(lib.f5_op_attention(P(qk), P(qkl), P(vt), P(vtl), P(ao), P(aol), P(None), B, H, N, npad, D, C.c_float(0.125), int(two), st))
            ck(lib.f5_op_gemm_resid_gate(P(ao), P(aol), P(blk["o"][0]), P(blk["o"][1]), P(blk["bo"]), P(W["ones"]), P(None), P(x), rows,
                                         D, D, D, D, D, ns, st))
            ck(lib.f5_op_ln_modulate(P(x), P(W["zeros"]), P(W["zeros"]), P(h), P(hl), rows, D, st))
            ck(lib.f5_op_gemm(P(h), P(hl), P(blk["ff1"][0]), P(blk["ff1"][1]), P(blk["bff1"]), P(None), P(ffh), P(ffhl), rows, T.ff_dim,
                              D, D, D, T.ff_dim, ns, 2, st))
            ck(lib.f5_op_gemm_resid_gate(P(ffh), P(ffhl), P(blk["ff2"][0]), P(blk["ff2"][1]), P(blk["bff2"]), P(W["ones"]), P(None),
                                         P(x), rows, D, T.ff_dim, T.ff_dim, T.ff_dim, D, ns, st))

        # ---- RMSNorm -> masked mean -> Linear(dim -> 1) -> Softplus (duration.py:137,188-190,249-251)
        pred = torch.empty((B,), device=dev)
        ck(lib.f5_op_duration_head(P(x), P(W["norm_out"]), P(W["to_pred"]), P(mask_d), P(pred), B, N, D, C.c_float(1e-5), st))
        return pred
