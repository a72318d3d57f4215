"""Round 6: hipGraph replay vs eager launches when EVERY call is followed by a host synchronisation (range_check "sync", the default; what
bench.py's headline and its b1_eager record measure) -- the start-up of a 5 084-node hipGraphLaunch is then not hidden behind the
previous call.  Modes: eager, one exec per call, one exec per ODE step (engine option graph_split).  Alternating legs, per-call wall
times, medians.
usage: python tools/r6_graph_sync_probe.py > gpurun_out/TAG/graph_sync_probe.jsonl"""
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
m = DiT.from_config(F5TTS_335M, precision="f16", device=dev)
m.load_weights(synthetic_weights(F5TTS_335M, seed=42))
f5 = F5TTS(transformer=m)
MODES = {"eager": dict(split=0, graph=False), "graph": dict(split=0, graph=True), "graph_split": dict(split=1, graph=True)}
cond, text, y0, _ = bench.synth_batch(1, 0, dev)
kw = dict(duration=bench.N_FRAMES, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
for check in ("sync", "off"):
    m.engine.range_check = check
    res = {k: [] for k in MODES}
    for k, v in MODES.items():
        m.engine.set_option("graph_split", v["split"])
        f5.sample(cond, text, use_graph=v["graph"], **kw)
        torch.cuda.synchronize()
    for rnd in range(5):
        for k, v in MODES.items():
            m.engine.set_option("graph_split", v["split"])
            for _ in range(8):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                f5.sample(cond, text, use_graph=v["graph"], **kw)
                torch.cuda.synchronize()
                res[k].append((time.perf_counter() - t0) * 1e3)
    print(json.dumps(dict(probe="graph_sync", range_check=check, per_call_sync=True,
                          median_ms={k: round(statistics.median(v), 3) for k, v in res.items()},
                          p10_ms={k: round(sorted(v)[len(v) // 10], 3) for k, v in res.items()})), flush=True)
m.engine.set_option("graph_split", 0)
