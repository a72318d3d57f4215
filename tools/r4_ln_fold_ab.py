"""A/B of the engine option "ln_fold" (LN-modulate folded into the epilogues of the block GEMMs, csrc/gemm.hpp fold_*) on bench.py's
workload: 32-point Euler sample(), f16, hipGraph replay, batch 4 / 8 / 16 / 32 -- ms per sample() with the option off and on, interleaved,
and the mel L1 between the two results.  One JSON line per batch size (profiles/r04/ln_fold_ab.jsonl).

    python tools/r4_ln_fold_ab.py [--batches 4,8,16,32] [--reps 3]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from f5_tts_mlx_amd.cfm import F5TTS  # noqa: E402
from f5_tts_mlx_amd.dit import DiT  # noqa: E402
from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batches", default="4,8,16,32")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--precision", default="f16")
    ap.add_argument("--only", type=int, default=None, help="run ONE option value (for rocprofv3 --kernel-trace --stats), eager, --steps points")
    ap.add_argument("--steps", type=int, default=32)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    m = DiT.from_config(F5TTS_335M, precision=a.precision, device=dev)
    m.load_weights(synthetic_weights(F5TTS_335M, seed=42))
    f5 = F5TTS(transformer=m)
    kw = dict(duration=bench.N_FRAMES, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, use_graph=True)
    if a.only is not None:
        B = int(a.batches.split(",")[0])
        cond, text, y0, _ = bench.synth_batch(B, 0, dev)
        m.engine.set_option("ln_fold", a.only)
        kw.update(steps=a.steps, use_graph=False)
        for _ in range(2):
            f5.sample(cond, text, y0=y0, **kw)
        torch.cuda.synchronize()
        return
    for B in [int(v) for v in a.batches.split(",")]:
        cond, text, y0, _ = bench.synth_batch(B, 0, dev)
        res, outs = {0: [], 1: []}, {}
        for rep in range(a.reps + 1):
            for opt in (0, 1):
                m.engine.set_option("ln_fold", opt)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out, _ = f5.sample(cond, text, y0=y0, **kw)
                torch.cuda.synchronize()
                if rep:                                   # rep 0 = capture
                    res[opt].append((time.perf_counter() - t0) * 1e3)
                outs[opt] = out
        m.engine.set_option("ln_fold", 0)
        off, on = min(res[0]), min(res[1])
        print(json.dumps(dict(B=B, precision=a.precision, ms_off=round(off, 2), ms_on=round(on, 2), speedup=round(off / on, 4),
                              all_off=[round(v, 2) for v in res[0]], all_on=[round(v, 2) for v in res[1]],
                              mel_l1_on_vs_off=float((outs[0] - outs[1]).abs().mean()))), flush=True)


if __name__ == "__main__":
    main()
