"""CPU ORACLE for the F5-TTS flow-matching sampling path.  TEST INFRASTRUCTURE ONLY.

This file restates, on torch-CPU / numpy, the algorithm of the reference
`lucasnewman/f5-tts-mlx` for the path `F5TTS.sample()` -> `DiT.__call__` (+ mel front-end).
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product path (f5_tts_mlx_amd/) never does.

PINNING STATUS: the reference ships no tests / golden vectors and its arithmetic lives in MLX
(`mlx>=0.18.1`, unpinned, pyproject.toml:35), which is not installable here.  What IS pinned: the
reference's own Python modules (dit, cfm, rope, convnext_v2, audio, utils, duration), imported
unmodified from /root/reference and executed over `oracle/mlx_shim.py` (a numpy emulation of the
mlx primitives they call), produce the committed vectors `tests/golden/ref_*.npz`
(`tests/golden/make_reference_golden.py`); this restatement reproduces them to fp32 rounding
(tests/test_reference_golden.py: forward 6e-6 max, trajectories < 1e-4 max, masks / tokens
bit-exact).  What is NOT pinned: MLX's own kernels — the meaning of each mlx primitive is the
shim's reading of the MLX documentation, never compared with MLX running ("parity unpinned" in
that sense).  Independent cross-checks (scipy STFT, torch.nn.functional) are in
tests/test_oracle.py.

Precision switches
  dtype            torch.float32 (reference arithmetic) or torch.float64 (ground truth)
  emulate_bf16     round GEMM/attention *operands* to bf16 at exactly the points where the HIP
                   engine does (fp32 accumulate).  Used to separate kernel bugs from bf16 drift.
  emulate_f16      the same with IEEE fp16 operands (the engine's "f16" precision).

All citations are file:line in /root/reference/f5_tts_mlx/.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# ------------------------------------------------------------------------------------------------
# host utilities (utils.py)
# ------------------------------------------------------------------------------------------------

def lens_to_mask(t: Tensor, length: Optional[int] = None) -> Tensor:
    """utils.py:39-47 — mask[b, n] = n < t[b]."""
    if length is None:
        length = int(t.max().item())
    seq = torch.arange(length)
    return seq[None, :] < t[:, None]


def pad_to_length(t: Tensor, length: int, value=0) -> Tensor:
    """utils.py:93-103."""
    seq_len = t.shape[-1]
    if length > seq_len:
        if t.ndim not in (1, 2):
            raise ValueError(f"Unsupported padding dims: {t.ndim}")
        t = F.pad(t, (0, length - seq_len), value=value)
    return t[..., :length]


def pad_sequence(ts: Sequence[Tensor], padding_value=0) -> Tensor:
    """utils.py:106-109."""
    max_len = max(int(i.shape[-1]) for i in ts)
    return torch.stack([pad_to_length(i, max_len, padding_value) for i in ts])


def list_str_to_tensor(text: List[str], padding_value=-1) -> Tensor:
    """utils.py:115-118 — utf-8 byte tokenizer, pad -1."""
    return pad_sequence([torch.tensor([*bytes(t, "UTF-8")], dtype=torch.int32) for t in text], padding_value=-1)


def list_str_to_idx(text: List[Union[str, List[str]]], vocab_char_map: Dict[str, int], padding_value=-1) -> Tensor:
    """utils.py:124-133 — vocab lookup, unknown -> 0, pad -1."""
    idx = [torch.tensor([vocab_char_map.get(c, 0) for c in t], dtype=torch.int32) for t in text]
    return pad_sequence(idx, padding_value=padding_value)


# ------------------------------------------------------------------------------------------------
# mel front-end (audio.py)
# ------------------------------------------------------------------------------------------------

def mel_filters(sample_rate: int, n_fft: int, n_mels: int, f_min: float = 0.0, f_max: Optional[float] = None) -> np.ndarray:
    """audio.py:12-98 with norm=None, mel_scale="htk".  Returns (n_mels, n_fft//2+1) float32.

    The reference evaluates linspace/arithmetics in float32 (MLX default); mirrored here.
    """
    def hz_to_mel(f):
        return 2595.0 * math.log10(1.0 + f / 700.0)

    f_max = f_max or sample_rate / 2
    n_freqs = n_fft // 2 + 1
    all_freqs = np.linspace(0, sample_rate // 2, n_freqs, dtype=np.float32)          # audio.py:71
    m_pts = np.linspace(hz_to_mel(f_min), hz_to_mel(f_max), n_mels + 2, dtype=np.float32)  # :75-77
    f_pts = (np.float32(700.0) * (np.float32(10.0) ** (m_pts / np.float32(2595.0)) - np.float32(1.0))).astype(np.float32)  # :54
    f_diff = f_pts[1:] - f_pts[:-1]                                                   # :82
    slopes = f_pts[None, :] - all_freqs[:, None]                                      # :83
    down = (-slopes[:, :-2]) / f_diff[:-1]                                            # :87
    up = slopes[:, 2:] / f_diff[1:]                                                   # :88
    fb = np.maximum(np.float32(0), np.minimum(down, up))                              # :89-91
    return np.ascontiguousarray(fb.T.astype(np.float32))                              # :97 moveaxis


def hanning(size: int) -> np.ndarray:
    """audio.py:101-112 — periodic Hann."""
    return np.hanning(size + 1)[:-1].astype(np.float32)


def stft(x: np.ndarray, window: np.ndarray, nperseg: int, noverlap: int, dtype=np.float32) -> np.ndarray:
    """audio.py:115-159 with pad_mode="constant": zero pad nperseg//2 both sides, frame count
    (len + noverlap - nperseg)... == (L + hop)//hop for nperseg=4*hop, rfft of windowed frames."""
    padding = nperseg // 2
    x = np.pad(x.astype(dtype), (padding, padding))
    t = (x.size - nperseg + noverlap) // noverlap
    # as_strided view of the padded signal (audio.py:155-158); indices stay in bounds because the
    # last frame starts at (t-1)*hop <= len - nperseg
    idx = np.arange(nperseg)[None, :] + noverlap * np.arange(t)[:, None]
    idx = np.minimum(idx, x.size - 1)  # defensive; never triggers for nperseg = 4*hop
    frames = x[idx] * window.astype(dtype)[None, :]
    spec = np.fft.rfft(frames.astype(np.float64 if dtype == np.float64 else np.float32), axis=-1)
    return spec


def log_mel_spectrogram(audio: np.ndarray, sample_rate=24_000, n_mels=100, n_fft=1024, hop_length=256,
                        dtype=np.float32) -> np.ndarray:
    """audio.py:162-210 — returns (b, frames, n_mels) (the code's layout, not the docstring's)."""
    audio = np.asarray(audio)
    if audio.ndim == 1:
        audio = audio[None]
    fb = mel_filters(sample_rate, n_fft, n_mels).astype(dtype)
    outs = []
    for i in range(audio.shape[0]):
        freqs = stft(audio[i], hanning(n_fft), nperseg=n_fft, noverlap=hop_length, dtype=dtype)
        mag = np.abs(freqs[:-1, :]).astype(dtype)            # audio.py:202 drops the last frame
        mel = mag @ fb.T                                     # :204
        outs.append(np.log(np.maximum(mel, dtype(1e-5))))    # :205
    return np.stack(outs, axis=0).astype(dtype)


# ------------------------------------------------------------------------------------------------
# activations with MLX semantics
# ------------------------------------------------------------------------------------------------

def _gelu_tanh(x: Tensor) -> Tensor:      # nn.GELU(approx="tanh") used by FeedForward (dit.py:94,309)
    return F.gelu(x, approximate="tanh")


def _gelu_erf(x: Tensor) -> Tensor:       # nn.GELU() in ConvNeXtV2Block (convnext_v2.py:42)
    return F.gelu(x)


def _mish(x: Tensor) -> Tensor:           # nn.Mish (dit.py:35,37)
    return x * torch.tanh(F.softplus(x))


def _bf16(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(x.dtype)


def _f16(x: Tensor) -> Tensor:
    """IEEE half rounding with the engine's saturation at +-65504 (csrc/op16.hpp f5_sat)."""
    return x.clamp(-65504.0, 65504.0).to(torch.float16).to(x.dtype)


# ------------------------------------------------------------------------------------------------
# positional tables (rope.py)
# ------------------------------------------------------------------------------------------------

def precompute_freqs_cis(dim: int, end: int, theta: float = 10000.0) -> Tensor:
    """rope.py:63-73 — [cos | sin] concatenated, float32."""
    freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].to(torch.float32) / dim))
    t = torch.arange(end, dtype=torch.float32)
    freqs = torch.outer(t, freqs).to(torch.float32)
    return torch.cat([freqs.cos(), freqs.sin()], dim=-1)


def get_pos_embed_indices(start: Tensor, length: int, max_pos: int, scale: float = 1.0) -> Tensor:
    """rope.py:76-84."""
    scale_t = scale * torch.ones_like(start, dtype=torch.float32)
    pos = start[:, None] + (torch.arange(length)[None, :] * scale_t[:, None]).to(torch.int32)
    return torch.where(pos < max_pos, pos, torch.full_like(pos, max_pos - 1))


def rotary_freqs(dim_head: int, seq_len: int, base: float = 10000.0) -> Tensor:
    """RotaryEmbedding.forward_from_seq_len (rope.py:12-60, xpos off): freqs (n, dim_head) with each
    inv_freq duplicated on interleaved pairs."""
    inv_freq = 1.0 / (base ** (torch.arange(0, dim_head, 2).to(torch.float32) / dim_head))
    t = torch.arange(seq_len).to(torch.float32)
    freqs = torch.einsum("i,j->ij", t, inv_freq)
    return torch.stack((freqs, freqs), dim=-1).reshape(seq_len, dim_head)


def rotate_half(x: Tensor) -> Tensor:
    """rope.py:87-91 — interleaved pairs (x0, x1) -> (-x1, x0)."""
    x = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = x[..., 0], x[..., 1]
    return torch.stack([-x2, x1], dim=-1).reshape(*x.shape[:-2], -1)


def apply_rotary_pos_emb(t: Tensor, freqs: Tensor, scale: float = 1.0) -> Tensor:
    """rope.py:94-107."""
    rot_dim, seq_len = freqs.shape[-1], t.shape[-2]
    freqs = freqs[-seq_len:, :].to(t.dtype)
    t_rot, t_pass = t[..., :rot_dim], t[..., rot_dim:]
    t_rot = (t_rot * freqs.cos() * scale) + (rotate_half(t_rot) * freqs.sin() * scale)
    return torch.cat((t_rot, t_pass), dim=-1)


# ------------------------------------------------------------------------------------------------
# DiT (dit.py, convnext_v2.py)
# ------------------------------------------------------------------------------------------------

class DiTOracle:
    """Restatement of `DiT` (dit.py:331-401) over a dict of reference-named fp32 weights.

    cfg: any object with attributes dim, depth, heads, dim_head, mel_dim, text_dim, conv_layers,
    conv_pos_kernel, conv_pos_groups, freq_embed_dim, text_max_pos (see weights.DiTConfig).
    """

    def __init__(self, cfg, weights: Dict[str, np.ndarray], dtype=torch.float32, emulate_bf16: bool = False,
                 emulate_mxfp8: bool = False, emulate_f16: bool = False, ln_fold: bool = False):
        """emulate_bf16: GEMM / attention operands rounded to bf16 like the engine's `bf16` mode.  emulate_mxfp8: the engine's
        `mxfp8` mode (BASELINE configs[4], no reference counterpart): as bf16, except that both operands of the four
        per-block linears (to_q/k/v, to_out, ff.0, ff.2) are MX-fp8 (oracle/mx_oracle.py), weights from their bf16 copies."""
        # ln_fold: restates the engine option of the same name (csrc/gemm.hpp fold_*): the LayerNorm-modulate steps of a block
        # are folded algebraically into the GEMM that consumes them -- the GEMM's A operand is the SHIFTED residual stream times
        # (1 + scale) rounded to 16 bits, the weight is the ordinary 16-bit W, and the normalisation arrives in the epilogue:
        #   (LN(x) (1 + s) + b) W^T + bias  =  rstd (((x - m)(1 + s)) W^T) - rstd (mu - m) c1 + c2,   c1 = W (1 + s),  c2 = W b + bias
        # with m = the row's mean at the PREVIOUS LayerNorm (round 5: the operand no longer carries the mean LayerNorm removes), mean /
        # rstd from the fp32 x, c1 / c2 fp32 sums over the rounded W.  Only meaningful together with an emulate_* rounding.  As in the
        # engine, block 0's first LN and the final one are ordinary LayerNorms (the first one supplies the first m).
        # ln_fold = "unshifted": m = 0 everywhere (the round-4 formulation, kept for the numerics study in tests / profiles/r05).
        self.ln_fold = ln_fold
        self._fold_m: Optional[Tensor] = None
        self.cfg = cfg
        self.dtype = dtype
        self.emu = emulate_bf16 or emulate_mxfp8 or emulate_f16
        self.mx = emulate_mxfp8
        self._r = _f16 if emulate_f16 else _bf16          # operand rounding of the emulated engine mode
        self.w = {k: torch.from_numpy(np.ascontiguousarray(v)).to(dtype) for k, v in weights.items()}
        self._wb: Dict[str, Tensor] = {}
        self.freqs_cis = precompute_freqs_cis(cfg.text_dim, cfg.text_max_pos).to(dtype)   # dit.py:191
        self.time_calls = 0

    # -- primitives ------------------------------------------------------------------------------
    def _W(self, name: str) -> Tensor:
        """GEMM weight operand (bf16-rounded under emulation)."""
        if not self.emu:
            return self.w[name]
        if name not in self._wb:
            self._wb[name] = self._r(self.w[name])
        return self._wb[name]

    def _A(self, x: Tensor) -> Tensor:
        """GEMM activation operand."""
        return self._r(x) if self.emu else x

    def linear(self, x: Tensor, name: str, lowp: bool = True) -> Tensor:
        """nn.Linear: x @ W.T + b, W (out, in)."""
        if lowp and self.mx and ".transformer_blocks." in name and (".attn.to_" in name or ".ff.ff." in name):
            from . import mx_oracle as MX
            key = name + ".weight#mx"
            if key not in self._wb:
                self._wb[key] = MX.mx_round(_bf16(self.w[name + ".weight"]).float()).to(self.dtype)
            return MX.mx_round(x.float()).to(self.dtype) @ self._wb[key].T + self.w[name + ".bias"]
        if lowp:
            return self._A(x) @ self._W(name + ".weight").T + self.w[name + ".bias"]
        return x @ self.w[name + ".weight"].T + self.w[name + ".bias"]

    @staticmethod
    def layer_norm(x: Tensor, weight: Optional[Tensor] = None, bias: Optional[Tensor] = None, eps: float = 1e-6) -> Tensor:
        """nn.LayerNorm — biased variance over the last axis."""
        mu = x.mean(dim=-1, keepdim=True)
        var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
        y = (x - mu) * torch.rsqrt(var + eps)
        if weight is not None:
            y = y * weight + bias
        return y

    def conv1d_cl(self, x: Tensor, name: str, groups: int, padding: int, lowp: bool) -> Tensor:
        """nn.Conv1d on channels-last input (b, n, c); weight (out, k, in/groups); cross-correlation,
        zero padding."""
        w = (self._W(name + ".weight") if lowp else self.w[name + ".weight"]).permute(0, 2, 1)  # (out, in/g, k)
        xin = self._A(x) if lowp else x
        y = F.conv1d(xin.transpose(1, 2), w, self.w[name + ".bias"], padding=padding, groups=groups)
        return y.transpose(1, 2)

    # -- modules ---------------------------------------------------------------------------------
    def time_embed(self, time: Tensor) -> Tensor:
        """TimestepEmbedding / SinusPositionEmbedding (dit.py:56-82).  fp32 in the engine (no bf16)."""
        half = self.cfg.freq_embed_dim // 2
        emb = math.log(10000) / (half - 1)
        emb = torch.exp(torch.arange(half, dtype=self.dtype) * -emb)
        emb = 1000 * time[:, None].to(self.dtype) * emb[None, :]
        emb = torch.cat([emb.sin(), emb.cos()], dim=-1)
        p = "transformer.time_embed.time_mlp.layers."
        h = F.silu(self.linear(emb, p + "0", lowp=False))
        return self.linear(h, p + "2", lowp=False)

    def convnext_block(self, x: Tensor, i: int) -> Tensor:
        """ConvNeXtV2Block + GRN (convnext_v2.py:9-54)."""
        p = f"transformer.text_embed.text_blocks.layers.{i}."
        residual = x
        x = self.conv1d_cl(x, p + "dwconv", groups=self.cfg.text_dim, padding=3, lowp=False)
        x = self.layer_norm(x, self.w[p + "norm.weight"], self.w[p + "norm.bias"], eps=1e-6)
        x = self.linear(x, p + "pwconv1")
        x = _gelu_erf(x)
        gx = torch.linalg.vector_norm(x, ord=2, dim=1, keepdim=True)          # over the SEQUENCE axis
        nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
        x = self.w[p + "grn.gamma"] * (x * nx) + self.w[p + "grn.beta"] + x
        x = self.linear(x, p + "pwconv2")
        return residual + x

    def text_embed(self, text: Tensor, seq_len: int, drop_text: bool) -> Tuple[Tensor, Tensor]:
        """TextEmbedding (dit.py:196-229). Returns (embedding (b,n,dt), ids actually embedded)."""
        batch, text_len = text.shape
        text = text.to(torch.int64) + 1                                       # dit.py:200
        text = text[:, :seq_len]                                              # :203
        text = F.pad(text, (0, seq_len - text_len), value=0)                  # :205
        text_mask = (text == 0)[..., None]                                    # :207 (before the drop)
        if drop_text:
            text = torch.zeros_like(text)                                     # :210
        ids = text
        x = self.w["transformer.text_embed.text_embed.weight"][text]          # :211
        if self.cfg.conv_layers > 0:
            pos_idx = get_pos_embed_indices(torch.zeros(batch, dtype=torch.int32), seq_len, self.cfg.text_max_pos)
            x = x + self.freqs_cis[pos_idx.to(torch.int64)]                   # :215-218
            mask_padding = bool(getattr(self.cfg, "text_mask_padding", True))      # :186; False -> plain Sequential (:227)
            if mask_padding:
                x = torch.where(text_mask, torch.zeros_like(x), x)            # :222
            for i in range(self.cfg.conv_layers):
                x = self.convnext_block(x, i)
                if mask_padding:
                    x = torch.where(text_mask, torch.zeros_like(x), x)        # :225
        return x, ids

    def conv_pos_embed(self, x: Tensor) -> Tensor:
        """ConvPositionEmbedding called WITHOUT mask (dit.py:29-50, :251)."""
        k, g = self.cfg.conv_pos_kernel, self.cfg.conv_pos_groups
        p = "transformer.input_embed.conv_pos_embed.conv1d.layers."
        x = _mish(self.conv1d_cl(x, p + "0", groups=g, padding=k // 2, lowp=True))
        x = _mish(self.conv1d_cl(x, p + "2", groups=g, padding=k // 2, lowp=True))
        return x

    def input_embed(self, x: Tensor, cond: Tensor, text_emb: Tensor, drop_audio_cond: bool) -> Tensor:
        """InputEmbedding (dit.py:241-252)."""
        if drop_audio_cond:
            cond = torch.zeros_like(cond)
        x = self.linear(torch.cat((x, cond, text_emb), dim=-1), "transformer.input_embed.proj")
        return self.conv_pos_embed(x) + x

    def fold_ln(self, x: Tensor, eps: float = 1e-6):
        """Row statistics of one folded LayerNorm (csrc/gemm.hip f5_fold_rows_kernel): d = x - m with m the row mean at the previous
        LayerNorm, mean and centred variance of d; leaves m := mean(x) for the next one.  Returns (d, rstd, mean_d)."""
        m = self._fold_m if (self._fold_m is not None and self.ln_fold != "unshifted") else torch.zeros_like(x[..., :1])
        d = x - m
        mean_d = d.mean(dim=-1, keepdim=True)
        rstd = torch.rsqrt(((d - mean_d) ** 2).mean(dim=-1, keepdim=True) + eps)
        self._fold_m = m + mean_d
        return d, rstd, mean_d

    def folded_linear(self, stats, scale: Tensor, shift: Tensor, name: str) -> Tensor:
        """(LN(x) (1 + scale) + shift) W^T + bias with the normalisation folded behind the GEMM (ln_fold, see __init__): stats =
        fold_ln(x) of the fp32 residual stream x (b, n, d), scale / shift (b, d)."""
        d, rstd, mean_d = stats
        w, bias = self.w[name + ".weight"], self.w[name + ".bias"]
        wr = self._r(w)                                                        # the ordinary weight operand
        acc = self._r(d * (1 + scale)[:, None, :]) @ wr.T                      # A operand = (x - m)(1 + scale), rounded
        c1 = (1 + scale) @ wr.T                                                # (b, out), fp32
        c2 = shift @ wr.T + bias                                               # (b, out), fp32
        return rstd * acc - (rstd * mean_d) * c1[:, None, :] + c2[:, None, :]

    def attention(self, x: Tensor, i: int, mask: Optional[Tensor], rope: Tensor, fold=None) -> Tensor:
        """Attention (dit.py:127-175).  fold = (fold_ln(x_residual), scale, shift): ln_fold, `x` is then unused for q / k / v."""
        p = f"transformer.transformer_blocks.{i}.attn."
        b, n, _ = x.shape
        H = self.cfg.heads
        if fold is not None:
            q, k, v = (self.folded_linear(fold[0], fold[1], fold[2], p + nm).reshape(b, n, H, -1).transpose(1, 2) for nm in ("to_q", "to_k", "to_v"))
        else:
            q = self.linear(x, p + "to_q").reshape(b, n, H, -1).transpose(1, 2)
            k = self.linear(x, p + "to_k").reshape(b, n, H, -1).transpose(1, 2)
            v = self.linear(x, p + "to_v").reshape(b, n, H, -1).transpose(1, 2)
        q = apply_rotary_pos_emb(q, rope)
        k = apply_rotary_pos_emb(k, rope)
        scale = 1.0 / math.sqrt(self.cfg.dim_head)
        q, k, v = self._A(q), self._A(k), self._A(v)
        s = (q @ k.transpose(-1, -2)) * scale
        if mask is not None:
            s = s.masked_fill(~mask[:, None, None, :], float("-inf"))        # bool mask, True = keep
        if self.emu:
            # engine: P = exp(s - max) rounded to bf16 for the PV product, row-sum kept in fp32
            m = s.amax(dim=-1, keepdim=True)
            pexp = torch.exp(s - m)
            o = (self._r(pexp) @ v) / pexp.sum(dim=-1, keepdim=True)
        else:
            o = torch.softmax(s, dim=-1) @ v
        o = o.transpose(1, 2).reshape(b, n, -1)
        o = self.linear(o, p + "to_out.layers.0")
        if mask is not None:
            o = o * mask[:, :, None].to(o.dtype)                              # dit.py:172-173
        return o

    def block(self, x: Tensor, t: Tensor, i: int, mask: Optional[Tensor], rope: Tensor) -> Tensor:
        """DiTBlock + AdaLayerNormZero (dit.py:259-325)."""
        p = f"transformer.transformer_blocks.{i}."
        emb = self.linear(F.silu(t), p + "attn_norm.linear", lowp=False)
        shift_msa, scale_msa, gate_msa, shift_mlp, scale_mlp, gate_mlp = emb.chunk(6, dim=1)
        if self.ln_fold and self.emu:
            if i == 0:                                   # the engine's LN kernel: an ordinary LayerNorm that also leaves the row means
                self._fold_m = x.mean(dim=-1, keepdim=True)
                norm = self.layer_norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
                x = x + gate_msa[:, None] * self.attention(norm, i, mask, rope)
            else:
                x = x + gate_msa[:, None] * self.attention(x, i, mask, rope, fold=(self.fold_ln(x), scale_msa, shift_msa))
            h = _gelu_tanh(self.folded_linear(self.fold_ln(x), scale_mlp, shift_mlp, p + "ff.ff.layers.0.layers.0"))
            ff = self.linear(h, p + "ff.ff.layers.2")
            return x + gate_mlp[:, None] * ff
        norm = self.layer_norm(x) * (1 + scale_msa[:, None]) + shift_msa[:, None]
        x = x + gate_msa[:, None] * self.attention(norm, i, mask, rope)
        norm = self.layer_norm(x) * (1 + scale_mlp[:, None]) + shift_mlp[:, None]
        h = _gelu_tanh(self.linear(norm, p + "ff.ff.layers.0.layers.0"))
        ff = self.linear(h, p + "ff.ff.layers.2")
        return x + gate_mlp[:, None] * ff

    def forward(self, x: Tensor, cond: Tensor, text: Tensor, time: Tensor, drop_audio_cond: bool, drop_text: bool,
                mask: Optional[Tensor] = None, return_hidden: bool = False):
        """DiT.__call__ (dit.py:374-401)."""
        x, cond = x.to(self.dtype), cond.to(self.dtype)
        batch, seq_len = x.shape[0], x.shape[1]
        time = torch.as_tensor(time, dtype=self.dtype)
        if time.ndim == 0:
            time = time.repeat(batch)
        t = self.time_embed(time)
        text_emb, _ = self.text_embed(text, seq_len, drop_text)
        x = self.input_embed(x, cond, text_emb, drop_audio_cond)
        rope = rotary_freqs(self.cfg.dim_head, seq_len)
        hidden = [x]
        for i in range(self.cfg.depth):
            x = self.block(x, t, i, mask, rope)
            if return_hidden:
                hidden.append(x)
        emb = self.linear(F.silu(t), "transformer.norm_out.linear", lowp=False)
        scale, shift = emb.chunk(2, dim=1)                                    # (scale, shift) order: dit.py:287
        x = self.layer_norm(x) * (1 + scale[:, None]) + shift[:, None]
        out = self.linear(x, "transformer.proj_out")
        return (out, hidden) if return_hidden else out


# ------------------------------------------------------------------------------------------------
# ODE solvers (cfm.py:38-122)
# ------------------------------------------------------------------------------------------------

def odeint_euler(func: Callable, y0: Tensor, t: Tensor) -> Tensor:
    ys, y = [y0], y0
    for i in range(len(t) - 1):
        dt = t[i + 1] - t[i]
        y = y + dt * func(t[i], y)
        ys.append(y)
    return torch.stack(ys)


def odeint_midpoint(func: Callable, y0: Tensor, t: Tensor) -> Tensor:
    ys, y = [y0], y0
    for i in range(len(t) - 1):
        dt = t[i + 1] - t[i]
        k1 = func(t[i], y)
        mid = y + 0.5 * dt * k1
        k2 = func(t[i] + 0.5 * dt, mid)
        y = y + dt * k2
        ys.append(y)
    return torch.stack(ys)


def odeint_rk4(func: Callable, y0: Tensor, t: Tensor) -> Tensor:
    ys, y = [y0], y0
    for i in range(len(t) - 1):
        dt = t[i + 1] - t[i]
        k1 = func(t[i], y)
        k2 = func(t[i] + 0.5 * dt, y + 0.5 * dt * k1)
        k3 = func(t[i] + 0.5 * dt, y + 0.5 * dt * k2)
        k4 = func(t[i] + dt, y + dt * k3)
        y = y + (dt / 6) * (k1 + 2 * k2 + 2 * k3 + k4)
        ys.append(y)
    return torch.stack(ys)


def time_grid(steps: int, sway_sampling_coef: Optional[float], dtype=torch.float32) -> Tensor:
    """cfm.py:377-381 — `steps` grid POINTS; sway warps them."""
    t = torch.linspace(0, 1, steps, dtype=dtype)
    if sway_sampling_coef is not None:
        t = t + sway_sampling_coef * (torch.cos(math.pi / 2 * t) - 1 + t)
    return t


# ------------------------------------------------------------------------------------------------
# F5TTS.sample (cfm.py:264-402)
# ------------------------------------------------------------------------------------------------

def sample(
    dit: DiTOracle,
    cond: Union[np.ndarray, Tensor],
    text: Union[Tensor, List[str], List[List[str]]],
    duration: Union[int, Tensor, None],
    *,
    lens: Optional[Tensor] = None,
    steps: int = 8,
    method: str = "rk4",
    cfg_strength: float = 2.0,
    sway_sampling_coef: Optional[float] = -1.0,
    max_duration: int = 4096,
    y0: Optional[Tensor] = None,
    vocab_char_map: Optional[Dict[str, int]] = None,
    vocoder: Optional[Callable] = None,
    return_aux: bool = False,
):
    """Restatement of `F5TTS.sample`.  Differences from the signature: the MLX PRNG is third-party,
    so the initial noise is INJECTED (`y0`, (b, n, d), zero beyond each element's duration) instead
    of `seed`; the duration predictor (out of scope) is not wired, so `duration=None` raises."""
    dtype = dit.dtype
    cond = torch.as_tensor(np.asarray(cond) if not isinstance(cond, Tensor) else cond)
    if cond.ndim == 2:                                                        # raw wave, batch 1 (cfm.py:283-286)
        assert cond.shape[0] == 1
        cond = torch.from_numpy(log_mel_spectrogram(cond[0].numpy().astype(np.float32)))
        assert cond.shape[-1] == dit.cfg.mel_dim
    cond = cond.to(dtype)
    batch, cond_seq_len = cond.shape[:2]
    if lens is None:
        lens = torch.full((batch,), cond_seq_len, dtype=torch.int64)          # float in the reference (:290)
    lens = torch.as_tensor(lens).to(torch.int64)

    if isinstance(text, list):                                                # :294-299
        text = list_str_to_idx(text, vocab_char_map) if vocab_char_map is not None else list_str_to_tensor(text)
        assert text.shape[0] == batch
    text = torch.as_tensor(text)
    text_lens = (text != -1).sum(dim=-1)                                      # :302
    lens = torch.maximum(text_lens, lens)                                     # :303

    if duration is None:
        raise ValueError("Duration must be provided or a duration predictor must be set.")   # :310
    cond_mask = lens_to_mask(lens)                                            # :312
    if isinstance(duration, int):
        duration = torch.full((batch,), duration, dtype=torch.int64)
    duration = torch.as_tensor(duration).to(torch.int64)
    duration = torch.maximum(lens + 1, duration)                              # :317
    duration = torch.clip(duration, 0, max_duration)                          # :318
    max_dur = int(duration.max().item())                                      # :319

    cond = F.pad(cond, (0, 0, 0, max_dur - cond_seq_len))                     # :321
    cond_mask = F.pad(cond_mask, (0, max_dur - cond_mask.shape[-1]), value=False)[..., None]
    step_cond = torch.where(cond_mask, cond, torch.zeros_like(cond))          # :331
    mask = lens_to_mask(duration) if batch > 1 else None                      # :333-336

    def fn(t, x):                                                             # :340-365
        pred = dit.forward(x, step_cond, text, t, False, False, mask)
        if cfg_strength < 1e-5:
            return pred
        null = dit.forward(x, step_cond, text, t, True, True, mask)
        return pred + (pred - null) * cfg_strength

    if y0 is None:
        raise ValueError("oracle.sample needs the initial noise y0 (MLX PRNG is not restated here)")
    y0 = torch.as_tensor(y0).to(dtype)
    assert y0.shape == (batch, max_dur, dit.cfg.mel_dim)

    t = time_grid(steps, sway_sampling_coef, dtype=torch.float32).to(dtype)
    if method == "midpoint":
        solver = odeint_midpoint
    elif method == "euler":
        solver = odeint_euler
    elif method == "rk4":
        solver = odeint_rk4
    else:
        raise ValueError(f"Unknown method: {method}")                         # :390

    trajectory = solver(fn, y0, t)                                            # :393
    out = torch.where(cond_mask, cond, trajectory[-1])                        # :395-397
    if vocoder is not None:
        out = vocoder(out)                                                    # :399-400
    if return_aux:
        return out, trajectory, dict(lens=lens, duration=duration, cond_mask=cond_mask[..., 0], mask=mask,
                                     step_cond=step_cond, t=t, text=text)
    return out, trajectory

# ------------------------------------------------------------------------------------------------
# F5TTS.__call__ — flow-matching training loss, forward only (cfm.py:169-251; utils.py:50-79)
# ------------------------------------------------------------------------------------------------

def mask_from_start_end_indices(seq_len: Tensor, start: Tensor, end: Tensor, max_length: int) -> Tensor:
    """utils.py:50-58: `max_length` is used unconditionally (the default to seq_len.max() is commented out)."""
    seq = torch.arange(max_length, dtype=torch.int32)
    return (seq[None, :] >= start[:, None]) & (seq[None, :] < end[:, None])


def mask_from_frac_lengths(seq_len: Tensor, frac_lengths: Tensor, rand: Tensor, max_length: int) -> Tensor:
    """utils.py:61-79 with the uniform draw `rand` (:68) injected.  float32 products truncated toward zero (`astype(int32)`)."""
    seq_len = torch.as_tensor(seq_len)
    lengths = (frac_lengths.to(torch.float32) * seq_len.to(torch.float32)).to(torch.int32)          # :65
    max_start = seq_len.to(torch.int32) - lengths                                                    # :66
    start = torch.clamp((max_start.to(torch.float32) * rand.to(torch.float32)).to(torch.int32), min=0)   # :70
    end = start + lengths                                                                            # :71
    out = mask_from_start_end_indices(seq_len, start, end, max_length)                               # :73
    return pad_to_length(out, max_length)                                                            # :75-76


def cfm_loss(
    dit: DiTOracle,
    inp: Tensor,                       # mel (b, n, d)
    text: Union[Tensor, List[str], List[List[str]]],
    *,
    lens: Optional[Tensor] = None,
    x0: Tensor,                        # (b, n, d)  the gaussian draw of cfm.py:201
    time: Tensor,                      # (b,)       uniform(0,1) draw of :204
    frac_lengths: Tensor,              # (b,)       uniform(*frac_lengths_mask) draw of :190
    span_rand: Tensor,                 # (b,)       uniform(0,1) draw inside mask_from_frac_lengths
    rand_audio_drop: float,            # :220
    rand_cond_drop: float,             # :221
    audio_drop_prob: float = 0.3,
    cond_drop_prob: float = 0.2,
    vocab_char_map: Optional[Dict[str, int]] = None,
    return_aux: bool = False,
):
    """Restatement of `F5TTS.__call__` with every random draw injected (the MLX PRNG is third-party).  Mel input only:
    the raw-wave branch (:177-180) transposes an already (b, frames, mels) array and is dead code in the reference."""
    dtype = dit.dtype
    inp = torch.as_tensor(inp).to(dtype)
    assert inp.ndim == 3 and inp.shape[-1] == dit.cfg.mel_dim
    batch, seq_len = inp.shape[:2]
    if isinstance(text, list):                                                # :185-190
        text = list_str_to_idx(text, vocab_char_map) if vocab_char_map is not None else list_str_to_tensor(text)
        assert text.shape[0] == batch
    text = torch.as_tensor(text)
    if lens is None:
        lens = torch.full((batch,), seq_len, dtype=torch.int32)               # :193-194
    lens = torch.as_tensor(lens)
    mask = lens_to_mask(lens, length=seq_len)                                 # :196
    rand_span_mask = mask_from_frac_lengths(lens, frac_lengths, span_rand, seq_len) & mask   # :199-203
    x1 = inp
    x0 = torch.as_tensor(x0).to(dtype)
    time = torch.as_tensor(time).to(dtype)
    t = time[:, None, None]
    phi = (1 - t) * x0 + t * x1                                               # :210
    flow = x1 - x0                                                            # :211
    cond = torch.where(rand_span_mask[..., None], torch.zeros_like(x1), x1)   # :214-218
    drop_text = bool(rand_cond_drop < cond_drop_prob)                         # :222-225
    drop_audio_cond = bool(rand_audio_drop < audio_drop_prob) or drop_text
    pred = dit.forward(phi, cond, text, time, drop_audio_cond, drop_text, None)   # :227-234 (no mask is passed)
    loss = (pred - flow) ** 2                                                 # :238
    m = rand_span_mask[..., None].expand(-1, -1, dit.cfg.mel_dim)             # :240
    masked = torch.where(m, loss, torch.zeros_like(loss))
    loss = masked.sum() / torch.clamp(m.sum().to(dtype), min=1e-6)            # :242
    if return_aux:
        return loss, dict(rand_span_mask=rand_span_mask, pred=pred, flow=flow, cond=cond, phi=phi,
                          drop_audio_cond=drop_audio_cond, drop_text=drop_text)
    return loss
