#!/usr/bin/env python
"""Golden vectors for the BATCHED full-size configurations (335M, N = 937, 5-point Euler + sway + CFG = 8 DiT forwards per
utterance), solved by the fp32 CPU oracle (oracle/f5_oracle.py, pinned to the reference's code by tests/test_reference_golden.py):

  full_b16_euler5.npz        bench.py's utterances 0..15 (SURVEY.md §8(d) seeds) as ONE batch of 16 equal durations (cfm.py:333-336:
                             the key-padding mask exists and keeps every key).  The oracle's batch elements do not interact at equal
                             durations, so the first B rows are also the oracle's answer for the batch of the first B utterances
                             (B = 2, 4, 8): one file serves the batch-2 ... 16 tests of the mid-size GEMM dispatch.
  full_b32_euler5.npz        bench.py's utterances 0..31 = BASELINE configs[2]'s batch (the configuration the roofline is quoted on) at the
                             same 5-point solve: rows 16..31 are solved here, rows 0..15 are taken from full_b16_euler5.npz (the
                             same oracle call on the same inputs; the elements of an equal-duration batch do not interact).
  full_b8_ragged_euler5.npz  a RAGGED batch of 8: durations 937 ... 500, text padded with -1 by a different amount per utterance
                             (cfm.py:317-336, dit.py:160-173: key mask, attention output rows zeroed at padded positions, GRN and
                             conv-pos over the padded length).

Inputs are regenerated from the seeds; only the oracle's final mels are stored, rounded to 17 significant bits (relative 4e-6:
two orders below the 1e-3 gate, and the file compresses to a third).  ~25 + 10 minutes of CPU on 8 cores:

    python tests/golden/make_batch_golden.py [--which b16|b32|ragged|all]     (b32: another ~25 minutes, needs the b16 file)
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import f5_oracle as O  # noqa: E402
from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights  # noqa: E402

N_FRAMES, REF_SAMPLES, NT, POINTS = 937, 72_000, 160, 5
RAGGED_DURATIONS = [937, 880, 811, 750, 689, 620, 561, 500]
KW = dict(steps=POINTS, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0)


def utterance(i: int):
    """bench.py synth_batch, utterance i, on the host: (wave, text ids, channel-major noise transposed to (N, 100))."""
    wave = np.random.default_rng(1234 + i).standard_normal(REF_SAMPLES).astype(np.float32) * np.float32(0.1)
    text = np.random.default_rng(2345 + i).integers(0, 2545, NT).astype(np.int32)
    y0 = np.ascontiguousarray(np.random.default_rng(3456 + i).standard_normal((100, N_FRAMES)).astype(np.float32).T)
    return wave, text, y0


def ragged_inputs():
    """The ragged batch: utterance i keeps its first RAGGED_DURATIONS[i] noise frames (zero beyond, cfm.py:374) and its first
    NT - 12 i text tokens (-1 beyond)."""
    B = len(RAGGED_DURATIONS)
    waves, text, y0 = [], np.full((B, NT), -1, np.int32), np.zeros((B, N_FRAMES, 100), np.float32)
    for i in range(B):
        w, t, y = utterance(i)
        waves.append(w)
        text[i, :NT - 12 * i] = t[:NT - 12 * i]
        y0[i, :RAGGED_DURATIONS[i]] = y[:RAGGED_DURATIONS[i]]
    return np.stack(waves), text, y0, np.asarray(RAGGED_DURATIONS, np.int64)


def round_mantissa(a: np.ndarray, keep_bits: int = 16) -> np.ndarray:
    """fp32 with the low (23 - keep_bits) mantissa bits rounded away (round to nearest)."""
    drop = 23 - keep_bits
    u = np.ascontiguousarray(a, np.float32).view(np.uint32).astype(np.uint64)
    u = ((u + (1 << (drop - 1))) >> drop) << drop
    return u.astype(np.uint32).view(np.float32).reshape(a.shape)


def mel_of(waves: np.ndarray) -> np.ndarray:
    return np.stack([np.asarray(O.log_mel_spectrogram(w), np.float32).reshape(-1, 100) for w in waves])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="all", choices=["b16", "b32", "ragged", "all"])
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ns = ap.parse_args()
    torch.set_num_threads(ns.threads)
    orc = O.DiTOracle(F5TTS_335M, synthetic_weights(F5TTS_335M, seed=42))
    if ns.which in ("b16", "all"):
        B = 16
        items = [utterance(i) for i in range(B)]
        cond = mel_of(np.stack([it[0] for it in items]))
        text = np.stack([it[1] for it in items])
        y0 = np.stack([it[2] for it in items])
        outs = []
        t0 = time.time()
        for c0 in range(0, B, 4):                      # chunks of 4: equal durations, the elements do not interact
            sl = slice(c0, c0 + 4)
            out, _ = O.sample(orc, torch.from_numpy(cond[sl]), torch.from_numpy(text[sl]), N_FRAMES, y0=torch.from_numpy(y0[sl]), **KW)
            outs.append(out.numpy().astype(np.float32))
            print(f"b16: utterances {c0}..{c0 + 3} done, {time.time() - t0:.0f} s", flush=True)
        out = np.concatenate(outs)
        np.savez_compressed(os.path.join(HERE, "full_b16_euler5.npz"), out=round_mantissa(out), points=np.int32(POINTS))
        print("wrote full_b16_euler5.npz", out.shape, float(np.abs(out).mean()))
    if ns.which in ("b32", "all"):
        first = np.load(os.path.join(HERE, "full_b16_euler5.npz"))["out"]
        items = [utterance(i) for i in range(16, 32)]
        cond = mel_of(np.stack([it[0] for it in items]))
        text = np.stack([it[1] for it in items])
        y0 = np.stack([it[2] for it in items])
        outs = [first]
        t0 = time.time()
        for c0 in range(0, 16, 4):
            sl = slice(c0, c0 + 4)
            out, _ = O.sample(orc, torch.from_numpy(cond[sl]), torch.from_numpy(text[sl]), N_FRAMES, y0=torch.from_numpy(y0[sl]), **KW)
            outs.append(round_mantissa(out.numpy().astype(np.float32)))
            print(f"b32: utterances {16 + c0}..{16 + c0 + 3} done, {time.time() - t0:.0f} s", flush=True)
        out = np.concatenate(outs)
        np.savez_compressed(os.path.join(HERE, "full_b32_euler5.npz"), out=out, points=np.int32(POINTS))
        print("wrote full_b32_euler5.npz", out.shape, float(np.abs(out).mean()))
    if ns.which in ("ragged", "all"):
        waves, text, y0, dur = ragged_inputs()
        cond = mel_of(waves)
        t0 = time.time()
        out, _, aux = O.sample(orc, torch.from_numpy(cond), torch.from_numpy(text), torch.from_numpy(dur), y0=torch.from_numpy(y0),
                               return_aux=True, **KW)
        print(f"ragged: {time.time() - t0:.0f} s; duration {aux['duration'].tolist()} lens {aux['lens'].tolist()}", flush=True)
        assert aux["duration"].tolist() == RAGGED_DURATIONS
        np.savez_compressed(os.path.join(HERE, "full_b8_ragged_euler5.npz"), out=round_mantissa(out.numpy().astype(np.float32)),
                            duration=aux["duration"].numpy().astype(np.int32), lens=aux["lens"].numpy().astype(np.int32),
                            points=np.int32(POINTS))
        print("wrote full_b8_ragged_euler5.npz", tuple(out.shape))


if __name__ == "__main__":
    main()
