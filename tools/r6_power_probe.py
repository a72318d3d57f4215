"""Round 6: is batch 32 bound by the schedule or by the power cap?  (GPU box, measurement only.)

The same sample() call -- same launches, same schedule -- with the real (synthetic N(0, 1/K)) weights and with the weights arena zeroed
(every MFMA operand 0: the matrix pipe toggles nothing), back-to-back calls for a few seconds each, per-call wall time and rocm-smi
clock / power samples while they run.  If the zero-operand run is faster at a higher clock and lower power, the real run is limited by
what the operand VALUES cost, not by idle slots.

usage: python tools/r6_power_probe.py [B ...] > gpurun_out/TAG/power_probe.jsonl
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
from tools.yardstick import SmiSampler  # noqa: E402

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
N = bench.N_FRAMES
m = DiT.from_config(F5TTS_335M, precision="f16", device=dev)
m.load_weights(synthetic_weights(F5TTS_335M, seed=42))
m.engine.range_check = "off"
f5 = F5TTS(transformer=m)
saved = m.engine.arena.clone()
smi = SmiSampler()


def leg(B, seconds):
    cond, text, y0, _ = bench.synth_batch(B, first=0, device=dev)
    kw = dict(duration=N, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0, use_graph=True)
    f5.sample(cond, text=text, **kw)
    torch.cuda.synchronize()
    ms = []
    import threading
    smi.samples, smi._stop = [], False
    th = threading.Thread(target=smi._run) if smi.ok else None
    if th:
        th.start()
    t_end = time.time() + seconds
    while time.time() < t_end:
        t0 = time.perf_counter()
        f5.sample(cond, text=text, **kw)
        torch.cuda.synchronize()
        ms.append(round((time.perf_counter() - t0) * 1e3, 2))
    smi._stop = True
    if th:
        th.join()
    return ms, smi.samples[2:]


for B in [int(x) for x in sys.argv[1:]] or [32, 1]:
    for what in ("real weights", "zeroed weights", "real weights"):
        if what.startswith("zero"):
            m.engine.arena.zero_()
        else:
            m.engine.arena.copy_(saved)
        torch.cuda.synchronize()
        ms, samples = leg(B, 8.0 if B >= 8 else 4.0)
        print(json.dumps(dict(kind="power_probe", batch=B, operands=what, calls=len(ms), first_ms=ms[0], median_ms=statistics.median(ms), last_ms=ms[-1],
                              smi_sclk_power=samples[:40])), flush=True)
