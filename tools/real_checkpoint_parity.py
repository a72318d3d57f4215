"""Parity and operand range of a REAL F5-TTS checkpoint (reference path cfm.py:404-520).

Every drift figure in this repo (f16 2.4e-4, bf16 1.9e-3) and the argument that IEEE-half operands are safe rest on seeded synthetic
weights -- no checkpoint is reachable offline.  Where one is, this tool answers the two open questions for it:

  1. mel L1 of the engine (f16 / bf16 / bf16x3) against the fp32 CPU oracle on the fixture WAV + its caption, short solve;
  2. the largest |value| that reaches each MFMA operand producer in the fp32 oracle run (LN-modulated activations into QKV / FF1,
     RoPE'd q and k, v, attention output into the out-projection, GELU output into FF2, conv-pos input, text-path pwconv inputs) --
     the f16 mode saturates at +-65 504 (op16.hpp f5_sat); anything within ~2x of that means: run this checkpoint in bf16x3 / bf16.

  3. (round 5) what the LN fold (engine option "ln_fold", default on from 22 000 rows) would meet: per LayerNorm of the fp32 oracle run
     the largest row mean in units of the row's sigma, the largest MOVE of a row mean since the previous LayerNorm (what the shifted
     operand still carries; DESIGN.md section 3: the folded error stays within 2x the unfolded one up to ~1 sigma of drift) and the largest
     |x - previous mean| (times |1 + scale| it must stay below 65 504 in f16); and, with --batch B >= 4, the engine with the fold
     FORCED ON next to the fold off on B copies of the utterance, with the count of operand-range fallbacks.

    python tools/real_checkpoint_parity.py /path/to/F5-TTS-dir [--steps 5] [--seconds 6.0] [--precisions f16,bf16,bf16x3] [--batch 4]

The directory is what F5TTS.from_pretrained reads: model_v1.safetensors (upstream or MLX key names) + vocab.txt.  The oracle is test
infrastructure: this tool is a checker, nothing on the product path imports it.  tests/test_model_gpu.py runs it on a synthetic
checkpoint directory.
"""
from __future__ import annotations

import argparse
import json
import os
import re
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

F16_MAX = 65504.0


def load_checkpoint(path: str):
    """-> (reference-named fp32 weights, vocab char map), the way F5TTS.from_pretrained reads a directory (cfm.py:411-421, 477-508)"""
    from safetensors.numpy import load_file
    from f5_tts_mlx_amd.weights import convert_upstream_weights
    vocab = {v: i for i, v in enumerate(open(os.path.join(path, "vocab.txt"), encoding="utf-8").read().split("\n"))}
    w = load_file(os.path.join(path, "model_v1.safetensors"))
    if any(k.startswith("ema_model.") for k in w):
        w = convert_upstream_weights(w)
    return {k: np.asarray(v, np.float32) for k, v in w.items()}, vocab


def recording_oracle(cfg, weights):
    """fp32 oracle that records max |x| per MFMA operand producer (keyed by site, block index stripped)"""
    from oracle import f5_oracle as O   # checker only

    class Rec(O.DiTOracle):
        def __init__(self, *a, **k):
            super().__init__(*a, **k)
            self.peaks, self._pending = {}, 0
            self.ln = dict(row_mean_over_sigma_max=0.0, mean_drift_over_sigma_max=0.0, shifted_abs_max=0.0, adaln_one_plus_scale_abs_max=0.0)
            self._prev_mean = None

        def layer_norm(self, x, weight=None, bias=None, eps=1e-6):
            if weight is None and x.shape[-1] == self.cfg.dim:      # the block / final adaLN LayerNorms of the residual stream
                mu = x.mean(-1, keepdim=True)
                sig = x.std(-1, unbiased=False, keepdim=True) + 1e-12
                self.ln["row_mean_over_sigma_max"] = max(self.ln["row_mean_over_sigma_max"], float((mu.abs() / sig).max()))
                if self._prev_mean is not None and self._prev_mean.shape == mu.shape:
                    self.ln["mean_drift_over_sigma_max"] = max(self.ln["mean_drift_over_sigma_max"], float(((mu - self._prev_mean).abs() / sig).max()))
                    self.ln["shifted_abs_max"] = max(self.ln["shifted_abs_max"], float((x - self._prev_mean).abs().max()))
                self._prev_mean = mu
            return O.DiTOracle.layer_norm(x, weight, bias, eps)

        def forward(self, *a, **k):
            self._prev_mean = None
            return super().forward(*a, **k)

        def _note(self, site, x):
            v = float(x.abs().max())
            self.peaks[site] = max(self.peaks.get(site, 0.0), v)

        def _A(self, x):
            if self._pending:                               # the three calls of attention(): q, k (after RoPE), v
                self._note(("attention q (RoPE'd, before the scale)", "attention k (RoPE'd)", "attention v")[3 - self._pending], x)
                self._pending -= 1
            return super()._A(x)

        def linear(self, x, name, lowp=True):
            if lowp:
                self._note("A operand of " + re.sub(r"\.\d+\.", ".N.", name), x)
            y = super().linear(x, name, lowp)
            if name.endswith("attn_norm.linear"):            # (shift, scale, gate) x 2: the scales are chunks 1 and 4
                ch = y.chunk(6, dim=1)
                self.ln["adaln_one_plus_scale_abs_max"] = max(self.ln["adaln_one_plus_scale_abs_max"], float((1 + ch[1]).abs().max()), float((1 + ch[4]).abs().max()))
            if name.endswith(".attn.to_v"):
                self._pending = 3
            return y

        def conv1d_cl(self, x, name, groups, padding, lowp):
            if lowp:
                self._note("conv input of " + name, x)
            return super().conv1d_cl(x, name, groups, padding, lowp)

    return Rec(cfg, weights)


def run(path: str, steps: int = 5, seconds: float = 6.0, precisions=("f16", "bf16", "bf16x3"), device: str = "cuda:0", wav: str | None = None,
        text: str = "Some call me nature, others call me mother nature.", batch: int = 1) -> dict:
    import dataclasses
    from oracle import f5_oracle as O   # checker only
    from f5_tts_mlx_amd.cfm import F5TTS
    from f5_tts_mlx_amd.dit import DiT
    from f5_tts_mlx_amd.generate import read_wav
    from f5_tts_mlx_amd.utils import convert_char_to_pinyin
    from f5_tts_mlx_amd.weights import F5TTS_335M
    weights, vocab = load_checkpoint(path)
    cfg = dataclasses.replace(F5TTS_335M, text_num_embeds=len(vocab) - 1)
    wav = wav or os.path.join(ROOT, "f5_tts_mlx_amd", "assets", "test_en_1_ref_short.wav")
    audio, sr = read_wav(wav)
    assert sr == 24000, "the reference audio must be 24 kHz (generate.py:147-148)"
    audio = np.asarray(audio, np.float32)
    n_ref = audio.shape[0] // 256
    duration = n_ref + int(seconds * 93.75)
    chars = convert_char_to_pinyin(["Some call me nature, others call me mother nature. " + text])
    r = np.random.default_rng(0)
    y0 = torch.from_numpy(np.ascontiguousarray(r.standard_normal((100, duration)).astype(np.float32).T))[None]
    kw = dict(steps=steps, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0)
    orc = recording_oracle(cfg, weights)
    cond = torch.from_numpy(np.asarray(O.log_mel_spectrogram(audio), np.float32).reshape(1, -1, 100))
    ref, _ = O.sample(orc, cond, chars, duration, y0=y0, vocab_char_map=vocab, **kw)
    res = dict(checkpoint=os.path.abspath(path), frames=int(duration), ref_frames=int(n_ref), ode_points=steps, forwards=2 * (steps - 1),
               oracle_mel_abs_mean=float(ref.abs().mean()), mel_l1={}, operand_peaks={k: round(v, 3) for k, v in sorted(orc.peaks.items())})
    worst = max(orc.peaks.values())
    res["layer_norm_statistics"] = {k: round(v, 4) for k, v in orc.ln.items()}
    res["ln_fold_operand_peak_estimate"] = orc.ln["shifted_abs_max"] * orc.ln["adaln_one_plus_scale_abs_max"]
    res["ln_fold_verdict"] = ("the folded operand (x - m)(1 + s) stays below 65504 / 8 and the row means move by <= 1 sigma between LayerNorms: "
                              "the fold is as accurate as the unfolded path here" if (res["ln_fold_operand_peak_estimate"] * 8 <= F16_MAX and orc.ln["mean_drift_over_sigma_max"] <= 1.0)
                              else "check the ln_fold lines below: large operand or fast-moving row means")
    res["largest_operand"] = worst
    res["f16_headroom"] = F16_MAX / worst
    res["verdict"] = ("f16 operands have >= 8x headroom on this input" if worst * 8 <= F16_MAX else
                      ("f16 operands are within 8x of saturation: compare the f16 and bf16x3 lines below" if worst <= F16_MAX else
                       "values beyond +-65504 reach MFMA operands: f16 saturates, use bf16x3 (or bf16)"))
    for prec in precisions:
        m = DiT.from_config(cfg, precision=prec, device=device)
        m.load_weights(weights)
        out, _ = F5TTS(transformer=m, vocab_char_map=vocab).sample(cond, chars, duration=duration, y0=y0, use_graph=False, **kw)
        torch.cuda.synchronize()
        res["mel_l1"][prec] = float((out.cpu() - ref).abs().mean())
        res.setdefault("finite", {})[prec] = bool(torch.isfinite(out).all())
        del m
        torch.cuda.empty_cache()
    if batch >= 4:
        # the LN fold needs the staged GEMM kernels (batch >= 4 at this size): B copies of the utterance, fold forced on vs off
        import warnings
        condb, y0b, charsb = cond.repeat(batch, 1, 1), y0.repeat(batch, 1, 1), chars * batch
        res["ln_fold"] = {}
        for prec in [p_ for p_ in precisions if p_ in ("f16", "bf16")]:
            m = DiT.from_config(cfg, precision=prec, device=device)
            m.load_weights(weights)
            m.engine.range_check = "sync"
            row = {}
            for opt in (0, 1):
                m.engine.set_option("ln_fold", opt)
                with warnings.catch_warnings(record=True) as wlist:
                    warnings.simplefilter("always")
                    try:
                        out, _ = F5TTS(transformer=m, vocab_char_map=vocab).sample(condb, charsb, duration=duration, y0=y0b, use_graph=False, **kw)
                    except RuntimeError as exc:                 # ln_fold = 1 where the block GEMMs do not run on the staged kernels
                        row["fold_on"] = dict(cannot_run=str(exc)[-160:], rows=2 * batch * int(duration))
                        continue
                    torch.cuda.synchronize()
                row["fold_on" if opt else "fold_off"] = dict(mel_l1_vs_oracle=float((out[0].cpu() - ref[0]).abs().mean()), finite=bool(torch.isfinite(out).all()),
                                                             fell_back=bool(wlist), ln_fold_after=m.engine.get_option("ln_fold"))
            row["operand_range_fallbacks"] = m.engine.range_events
            res["ln_fold"][prec] = row
            del m
            torch.cuda.empty_cache()
    return res


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("path")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--seconds", type=float, default=6.0)
    ap.add_argument("--precisions", default="f16,bf16,bf16x3")
    ap.add_argument("--wav", default=None)
    ap.add_argument("--batch", type=int, default=1, help=">= 4: also run B copies with the LN fold forced on / off (engine option ln_fold)")
    a = ap.parse_args()
    print(json.dumps(run(a.path, a.steps, a.seconds, tuple(a.precisions.split(",")), wav=a.wav, batch=a.batch), indent=1))
