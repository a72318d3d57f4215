"""Round 6: sample() unsplit vs split into two half batches on two streams (Engine.split_batch), the shipped implementation, by batch
size.  Alternating legs of back-to-back calls, per-call wall times (the clock follows the power of the last seconds: the first call
after a pause is 5 % faster than the sustained ones, so medians of legs are compared, not best-of).

usage: python tools/r6_split_ab.py [B ...] > gpurun_out/TAG/split_ab.jsonl
"""
import json
import os
import statistics
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from f5_tts_mlx_amd.cfm import F5TTS  # noqa: E402
from f5_tts_mlx_amd.dit import DiT  # noqa: E402
from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
N = bench.N_FRAMES
m = DiT.from_config(F5TTS_335M, precision="f16", device=dev)
m.load_weights(synthetic_weights(F5TTS_335M, seed=42))
f5 = F5TTS(transformer=m)
eng = m.engine


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


for B in [int(x) for x in sys.argv[1:]] or [16, 32, 64]:
    cond, text, y0, _ = bench.synth_batch(B, first=0, device=dev)
    kw = dict(duration=N, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0, use_graph=True)
    call = lambda: f5.sample(cond, text=text, **kw)   # noqa: E731
    outs = {}
    for sb in (0, B):                                  # capture both
        eng.split_batch = sb
        outs[sb] = call()[0].clone()
        call()
    rec = dict(kind="split_ab", batch=B, frames=N, bit_identical=bool(torch.equal(outs[0], outs[B])), legs=[])
    calls = 4 if B <= 32 else 3
    for leg in range(6):
        sb = 0 if leg % 2 == 0 else B
        eng.split_batch = sb
        rec["legs"].append({"split" if sb else "unsplit": [round(wall(call), 1) for _ in range(calls)]})
    med = lambda k: statistics.median(x for leg in rec["legs"] for kk, v in leg.items() if kk == k for x in v[1:])   # noqa: E731
    rec["median_ms"] = {"unsplit": med("unsplit"), "split": med("split")}
    rec["split_vs_unsplit"] = round(rec["median_ms"]["split"] / rec["median_ms"]["unsplit"], 4)
    print(json.dumps(rec), flush=True)
