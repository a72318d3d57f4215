"""Parameter inventory, synthetic weights and checkpoint key conversion for the F5-TTS DiT.

The names/shapes are the ones the reference model exposes after `F5TTS.load_weights`
(reference: f5_tts_mlx/cfm.py:459-517, f5_tts_mlx/dit.py:331-372, f5_tts_mlx/convnext_v2.py:24-44;
SURVEY.md Appendix B).  Layout is the reference's (MLX) layout:

  * Linear weight  (out, in)
  * Conv1d weight  (out, k, in/groups)          (cfm.py:499-504 converts torch (out,in/g,k) -> this)
  * Embedding      (num, dim)

There is no network in the build environment, hence no real checkpoint; `synthetic_weights`
draws seeded weights with the distribution stated in SURVEY.md §8(d) so that engine, oracle,
tests and bench all see the same tensors.
"""
from __future__ import annotations

from dataclasses import dataclass, asdict
from typing import Dict, Iterable, List, Tuple

import numpy as np


@dataclass(frozen=True)
class DiTConfig:
    """Constructor arguments of the reference `DiT` (dit.py:332-346)."""

    dim: int = 1024
    depth: int = 22
    heads: int = 16
    dim_head: int = 64
    ff_mult: int = 2
    mel_dim: int = 100
    text_num_embeds: int = 2545
    text_dim: int = 512
    conv_layers: int = 4
    conv_mult: int = 2           # TextEmbedding conv_mult (dit.py:182)
    conv_pos_kernel: int = 31    # ConvPositionEmbedding (dit.py:30)
    conv_pos_groups: int = 16
    freq_embed_dim: int = 256    # TimestepEmbedding (dit.py:74)
    text_max_pos: int = 4096     # TextEmbedding.precompute_max_pos (dit.py:190)
    text_mask_padding: bool = True   # TextEmbedding mask_padding (dit.py:182,186): zero filler / padded text positions

    @property
    def ff_dim(self) -> int:
        return int(self.dim * self.ff_mult)

    @property
    def text_ff_dim(self) -> int:
        return self.text_dim * self.conv_mult

    def as_dict(self):
        return asdict(self)


F5TTS_335M = DiTConfig()  # cfm.py:459-469

# a small config used by the CPU/GPU parity tests (oracle finishes in seconds)
TINY = DiTConfig(dim=256, depth=2, heads=4, dim_head=64, ff_mult=2, mel_dim=100,
                 text_num_embeds=64, text_dim=256, conv_layers=2, conv_pos_groups=4)


def param_specs(cfg: DiTConfig) -> List[Tuple[str, Tuple[int, ...], str]]:
    """(name, shape, kind) in the canonical order used for seeded generation."""
    D, Dt, FF, TF = cfg.dim, cfg.text_dim, cfg.ff_dim, cfg.text_ff_dim
    gin = cfg.dim // cfg.conv_pos_groups
    specs: List[Tuple[str, Tuple[int, ...], str]] = []
    add = specs.append
    p = "transformer."
    add((p + "time_embed.time_mlp.layers.0.weight", (D, cfg.freq_embed_dim), "linear"))
    add((p + "time_embed.time_mlp.layers.0.bias", (D,), "bias"))
    add((p + "time_embed.time_mlp.layers.2.weight", (D, D), "linear"))
    add((p + "time_embed.time_mlp.layers.2.bias", (D,), "bias"))
    add((p + "text_embed.text_embed.weight", (cfg.text_num_embeds + 1, Dt), "embedding"))
    for i in range(cfg.conv_layers):
        q = f"{p}text_embed.text_blocks.layers.{i}."
        add((q + "dwconv.weight", (Dt, 7, 1), "conv"))
        add((q + "dwconv.bias", (Dt,), "bias"))
        add((q + "norm.weight", (Dt,), "ln_weight"))
        add((q + "norm.bias", (Dt,), "bias"))
        add((q + "pwconv1.weight", (TF, Dt), "linear"))
        add((q + "pwconv1.bias", (TF,), "bias"))
        add((q + "grn.gamma", (1, 1, TF), "grn"))
        add((q + "grn.beta", (1, 1, TF), "grn"))
        add((q + "pwconv2.weight", (Dt, TF), "linear"))
        add((q + "pwconv2.bias", (Dt,), "bias"))
    add((p + "input_embed.proj.weight", (D, 2 * cfg.mel_dim + Dt), "linear"))
    add((p + "input_embed.proj.bias", (D,), "bias"))
    for j in (0, 2):
        add((f"{p}input_embed.conv_pos_embed.conv1d.layers.{j}.weight", (D, cfg.conv_pos_kernel, gin), "conv"))
        add((f"{p}input_embed.conv_pos_embed.conv1d.layers.{j}.bias", (D,), "bias"))
    for i in range(cfg.depth):
        q = f"{p}transformer_blocks.{i}."
        add((q + "attn_norm.linear.weight", (6 * D, D), "adaln"))
        add((q + "attn_norm.linear.bias", (6 * D,), "bias"))
        for nm in ("to_q", "to_k", "to_v"):
            add((q + f"attn.{nm}.weight", (D, D), "linear"))
            add((q + f"attn.{nm}.bias", (D,), "bias"))
        add((q + "attn.to_out.layers.0.weight", (D, D), "linear"))
        add((q + "attn.to_out.layers.0.bias", (D,), "bias"))
        add((q + "ff.ff.layers.0.layers.0.weight", (FF, D), "linear"))
        add((q + "ff.ff.layers.0.layers.0.bias", (FF,), "bias"))
        add((q + "ff.ff.layers.2.weight", (D, FF), "linear"))
        add((q + "ff.ff.layers.2.bias", (D,), "bias"))
    add((p + "norm_out.linear.weight", (2 * D, D), "adaln"))
    add((p + "norm_out.linear.bias", (2 * D,), "bias"))
    add((p + "proj_out.weight", (cfg.mel_dim, D), "linear"))
    add((p + "proj_out.bias", (cfg.mel_dim,), "bias"))
    return specs


def num_params(cfg: DiTConfig) -> int:
    return int(sum(int(np.prod(s)) for _, s, _ in param_specs(cfg)))


def synthetic_weights(cfg: DiTConfig = F5TTS_335M, seed: int = 42) -> Dict[str, np.ndarray]:
    """Seeded fp32 weights, distribution of SURVEY.md §8(d):

    Linear/Conv ~ N(0, 1/fan_in); biases ~ N(0, 0.02^2); adaLN linears ~ N(0, 0.25/fan_in);
    Embedding ~ N(0,1); GRN gamma/beta ~ N(0, 0.1^2); ConvNeXt LN weight = 1 + N(0, 0.02^2).
    Drawn with numpy `default_rng(seed)` in `param_specs` order.
    """
    rng = np.random.default_rng(seed)
    out: Dict[str, np.ndarray] = {}
    for name, shape, kind in param_specs(cfg):
        if kind == "linear":
            std = (1.0 / shape[1]) ** 0.5
        elif kind == "conv":
            std = (1.0 / (shape[1] * shape[2])) ** 0.5
        elif kind == "adaln":
            std = (0.25 / shape[1]) ** 0.5
        elif kind == "bias":
            std = 0.02
        elif kind == "embedding":
            std = 1.0
        elif kind == "grn":
            std = 0.1
        elif kind == "ln_weight":
            w = 1.0 + 0.02 * rng.standard_normal(shape, dtype=np.float32)
            out[name] = w.astype(np.float32)
            continue
        else:  # pragma: no cover
            raise ValueError(kind)
        out[name] = (rng.standard_normal(shape, dtype=np.float32) * np.float32(std)).astype(np.float32)
    return out


# ----------------------------------------------------------------------------------------------
# upstream (PyTorch F5-TTS) checkpoint -> reference key/layout conversion  (cfm.py:477-508)
# ----------------------------------------------------------------------------------------------

def convert_upstream_weights(weights: Dict[str, np.ndarray]) -> Dict[str, np.ndarray]:
    """Apply the renames / axis swaps `F5TTS.from_pretrained(convert_weights=True)` applies.

    Same observable mapping as cfm.py:479-506 (first matching rename wins, like the elif chain).
    """
    renames = (
        (".to_out", ".to_out.layers"),
        (".text_blocks", ".text_blocks.layers"),
        (".ff.ff.0.0", ".ff.ff.layers.0.layers.0"),
        (".ff.ff.2", ".ff.ff.layers.2"),
        (".time_mlp", ".time_mlp.layers"),
        (".conv1d", ".conv1d.layers"),
    )
    out: Dict[str, np.ndarray] = {}
    for k, v in weights.items():
        k = k.replace("ema_model.", "")
        if len(k) < 1 or "mel_spec." in k or k in ("initted", "step"):
            continue
        for old, new in renames:
            if old in k:
                k = k.replace(old, new)
                break
        if ".dwconv.weight" in k or ".conv1d.layers.0.weight" in k or ".conv1d.layers.2.weight" in k:
            v = np.swapaxes(np.asarray(v), 1, 2)
        out[k] = np.ascontiguousarray(v)
    return out


# ----------------------------------------------------------------------------------------------
# MLX group-quantized checkpoints (`model_v1_{4,8}b.safetensors`, cfm.py:450-452, 510-515)
# ----------------------------------------------------------------------------------------------
# `nn.quantize(bits, group_size=64)` replaces every Linear whose input width is a multiple of 64 by (weight: uint32
# words, scales, biases).  MLX's affine scheme (mlx docs, `mx.quantize`; third-party, NOT verifiable in this container):
# per group of 64 consecutive input elements  w ~= scale * q + bias  with q in [0, 2^bits), bias = group minimum and
# scale = (max - min) / (2^bits - 1); the q of 32/bits consecutive elements are packed into one uint32, the first
# element in the LEAST significant bits.  The engine has no int4/int8 GEMM: such checkpoints are expanded to fp32 here
# and then run through the same bf16 / bf16x3 path as a full-precision checkpoint.

def quantize_mlx_affine(w: np.ndarray, bits: int, group_size: int = 64):
    """Restatement of MLX's affine group quantisation (used by the tests and by tools that write such checkpoints)."""
    assert bits in (2, 4, 8) and w.ndim == 2 and w.shape[1] % group_size == 0
    out_f, in_f = w.shape
    g = np.asarray(w, np.float32).reshape(out_f, in_f // group_size, group_size)
    lo, hi = g.min(axis=-1, keepdims=True), g.max(axis=-1, keepdims=True)
    levels = float((1 << bits) - 1)
    scale = np.maximum((hi - lo) / levels, 1e-7).astype(np.float32)
    q = np.clip(np.rint((g - lo) / scale), 0, levels).astype(np.uint32).reshape(out_f, in_f)
    per = 32 // bits
    packed = np.zeros((out_f, in_f // per), np.uint32)
    for i in range(per):
        packed |= q[:, i::per] << np.uint32(bits * i)
    return packed, scale[..., 0], lo[..., 0].astype(np.float32)


def dequantize_mlx_affine(packed: np.ndarray, scales: np.ndarray, biases: np.ndarray, bits: int, group_size: int = 64) -> np.ndarray:
    packed = np.asarray(packed).astype(np.uint32)
    per = 32 // bits
    out_f, words = packed.shape
    in_f = words * per
    q = np.empty((out_f, in_f), np.float32)
    mask = np.uint32((1 << bits) - 1)
    for i in range(per):
        q[:, i::per] = ((packed >> np.uint32(bits * i)) & mask).astype(np.float32)
    s = np.repeat(np.asarray(scales, np.float32), group_size, axis=1)
    b = np.repeat(np.asarray(biases, np.float32), group_size, axis=1)
    if s.shape != q.shape:
        raise ValueError(f"quantised tensor {packed.shape} does not match scales {np.shape(scales)} at {bits} bits")
    return q * s + b


def dequantize_mlx_checkpoint(weights: Dict[str, np.ndarray], bits: int, group_size: int = 64) -> Dict[str, np.ndarray]:
    """Expand every (X.weight uint32, X.scales, X.biases) triple to a dense fp32 X.weight; other tensors pass through."""
    out: Dict[str, np.ndarray] = {}
    for k, v in weights.items():
        if k.endswith(".scales") or k.endswith(".biases"):
            continue
        base = k[: -len(".weight")] if k.endswith(".weight") else None
        if base is not None and base + ".scales" in weights:
            out[k] = dequantize_mlx_affine(v, weights[base + ".scales"], weights[base + ".biases"], bits, group_size)
        else:
            out[k] = v
    return out


def check_weights(cfg: DiTConfig, weights: Dict[str, np.ndarray], ignore: Iterable[str] = ("transformer.rotary_embed.inv_freq",)) -> None:
    """Raise ValueError when `weights` does not match the parameter inventory of `cfg`."""
    want = {n: s for n, s, _ in param_specs(cfg)}
    for n, s in want.items():
        if n not in weights:
            raise ValueError(f"missing parameter {n}")
        if tuple(weights[n].shape) != tuple(s):
            raise ValueError(f"shape mismatch for {n}: got {tuple(weights[n].shape)}, want {tuple(s)}")
    extra = [k for k in weights if k not in want and k not in set(ignore)]
    if extra:
        raise ValueError(f"unexpected parameters: {extra[:5]}{'...' if len(extra) > 5 else ''}")
