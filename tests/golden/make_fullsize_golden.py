#!/usr/bin/env python
"""Golden vector of the BENCHMARK workload at full depth (BASELINE configs[1]): F5-TTS 335M, one 10 s utterance (N = 937),
32-point Euler with sway sampling and CFG 2.0 = 62 DiT forwards, run through the fp32 CPU oracle (oracle/f5_oracle.py,
the restatement of cfm.py:264-402 / dit.py:374-401 pinned by tests/test_reference_golden.py).

The inputs are bench.py's utterance 0 (SURVEY.md §8(d) seeds): reference audio default_rng(1234), text default_rng(2345),
initial noise default_rng(3456), weights synthetic_weights(F5TTS_335M, seed=42); they are regenerated from the seeds, only
the oracle's answer is stored.  Takes ~5-10 minutes of CPU (27.4 TFLOP); the GPU box only reads the .npz:

    python tests/golden/make_fullsize_golden.py            # writes tests/golden/full_b1_euler32.npz (~0.9 MB)
    python tests/golden/make_fullsize_golden.py --emulate f16   # prints the drift the fp16-operand mode should show
    python tests/golden/make_fullsize_golden.py --method midpoint --points 16   # BASELINE configs[4]'s solver on the same utterance:
                                                                               # writes tests/golden/full_b1_midpoint16.npz (`out` only)
    python tests/golden/make_fullsize_golden.py --method midpoint --points 16 --emulate mxfp8   # expected drift of the mxfp8 mode

Stored: `out` = the final mel where(cond_mask, cond, x1) (937, 100) fp32, `traj_8/16/24` = trajectory states after 8 / 16 / 24
updates (drift by depth), `cond281` = the oracle's mel front-end output for the reference audio (281, 100).
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

N_FRAMES, REF_SAMPLES, NT, ODE_POINTS = 937, 72_000, 160, 32
OUT = os.path.join(HERE, "full_b1_euler32.npz")


def inputs(i: int = 0):
    """bench.py synth_batch, utterance i, on the host."""
    wave = np.random.default_rng(1234 + i).standard_normal(REF_SAMPLES).astype(np.float32) * np.float32(0.1)
    text = np.random.default_rng(2345 + i).integers(0, 2545, NT).astype(np.int32)
    y0 = np.ascontiguousarray(np.random.default_rng(3456 + i).standard_normal((100, N_FRAMES)).astype(np.float32).T)
    return wave, text, y0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--emulate", default=None, choices=[None, "bf16", "f16", "mxfp8"], help="report operand-rounding drift instead of writing")
    ap.add_argument("--method", default="euler", choices=["euler", "midpoint", "rk4"])
    ap.add_argument("--points", type=int, default=ODE_POINTS)
    ap.add_argument("--ln-fold", action="store_true", help="numerics study: LN-modulate folded algebraically into the consumer GEMMs (oracle ln_fold)")
    ap.add_argument("--threads", type=int, default=os.cpu_count())
    ns = ap.parse_args()
    torch.set_num_threads(ns.threads)
    wave, text, y0 = inputs(0)
    cond = np.asarray(O.log_mel_spectrogram(wave), dtype=np.float32).reshape(-1, 100)   # (281, 100)
    assert cond.shape == (REF_SAMPLES // 256, 100)
    w = synthetic_weights(F5TTS_335M, seed=42)
    kw = {} if ns.emulate is None else {f"emulate_{ns.emulate}": True}
    if ns.ln_fold:
        kw["ln_fold"] = True
    orc = O.DiTOracle(F5TTS_335M, w, **kw)
    t0 = time.time()
    out, traj = O.sample(orc, torch.from_numpy(cond)[None], torch.from_numpy(text)[None], N_FRAMES, y0=torch.from_numpy(y0)[None],
                         steps=ns.points, method=ns.method, cfg_strength=2.0, sway_sampling_coef=-1.0)
    print(f"oracle sample: {time.time() - t0:.1f} s, out {tuple(out.shape)}, |out| mean {float(out.abs().mean()):.4f}")
    if (ns.method, ns.points) != ("euler", ODE_POINTS):
        alt = os.path.join(HERE, f"full_b1_{ns.method}{ns.points}.npz")
        if ns.emulate is None:
            np.savez_compressed(alt, out=out[0].numpy().astype(np.float32))
            print("wrote", alt, os.path.getsize(alt), "bytes")
        else:
            g = np.load(alt)
            print(f"[{ns.emulate}{' + ln_fold' if ns.ln_fold else ''}] {ns.points}-point {ns.method}: mel L1 vs fp32 oracle, out: "
                  f"{float(np.abs(out[0].numpy() - g['out']).mean()):.3e}")
        return
    if ns.emulate is None:
        np.savez_compressed(OUT, out=out[0].numpy().astype(np.float32), traj_8=traj[8, 0].numpy().astype(np.float32),
                            traj_16=traj[16, 0].numpy().astype(np.float32), traj_24=traj[24, 0].numpy().astype(np.float32),
                            cond281=cond.astype(np.float32))
        print("wrote", OUT, os.path.getsize(OUT), "bytes")
    else:
        g = np.load(OUT)
        for k, v in (("out", out[0]), ("traj_8", traj[8, 0]), ("traj_16", traj[16, 0]), ("traj_24", traj[24, 0])):
            print(f"[{ns.emulate}{' + ln_fold' if ns.ln_fold else ''}] mel L1 vs fp32 oracle, {k}: {float(np.abs(v.numpy() - g[k]).mean()):.3e}")


if __name__ == "__main__":
    main()
