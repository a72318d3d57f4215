"""Round 6 (VERDICT r5 "next" #5): why does a 5 084-node hipGraph replay 2 ms slower than eager launches of the same kernels?
The same sample() (335M, 32-point Euler, batch 1 / 32) three ways, interleaved round by round on one box:
  eager        use_graph=False: ~5 100 launches from the host thread
  graph        one hipGraphExec for the whole call (the shipped graph mode)
  graph_split  engine option graph_split = 1: one exec per ODE step (prep rides in the first), 31 hipGraphLaunch calls back to back --
               the same kernels with the same arguments
Prints, per mode: ms per call (wall clock around `reps` calls, one synchronisation at the end), host ms per call until sample() returns
(range_check "off": the call does not block), and whether the outputs are bit-identical.
usage: python tools/r6_graph_probe.py [batch ...] > profiles/r06/graph_split_probe.jsonl"""
import json
import os
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
m.engine.range_check = "off"
f5 = F5TTS(transformer=m)
MODES = {"eager": dict(split=0, graph=False), "graph": dict(split=0, graph=True), "graph_split": dict(split=1, graph=True)}
for B in [int(x) for x in (sys.argv[1:] or ["1", "32"])]:
    cond, text, y0, _ = bench.synth_batch(B, 0, dev)
    kw = dict(duration=bench.N_FRAMES, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
    reps = 6 if B == 1 else 3
    outs, res = {}, {k: dict(ms=[], host_ms=[]) for k in MODES}
    for k, v in MODES.items():                                   # capture / warm every mode once
        m.engine.set_option("graph_split", v["split"])
        t0 = time.perf_counter()
        o, _ = f5.sample(cond, text, use_graph=v["graph"], **kw)
        torch.cuda.synchronize()
        res[k]["first_call_ms"] = (time.perf_counter() - t0) * 1e3
        outs[k] = o.clone()
    for rnd in range(5):
        for k, v in MODES.items():
            m.engine.set_option("graph_split", v["split"])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            host = 0.0
            for _ in range(reps):
                h0 = time.perf_counter()
                o, _ = f5.sample(cond, text, use_graph=v["graph"], **kw)
                host += time.perf_counter() - h0
            torch.cuda.synchronize()
            res[k]["ms"].append((time.perf_counter() - t0) / reps * 1e3)
            res[k]["host_ms"].append(host / reps * 1e3)
    rec = dict(probe="graph_split", batch=B, reps_per_round=reps, graphs_cached=m.engine.graph_count(),
               bit_identical=bool(torch.equal(outs["eager"], outs["graph"]) and torch.equal(outs["eager"], outs["graph_split"])))
    for k in MODES:
        rec[k] = dict(ms_min=round(min(res[k]["ms"]), 3), ms_median=round(sorted(res[k]["ms"])[2], 3), ms_all=[round(x, 2) for x in res[k]["ms"]],
                      host_ms_per_call=round(min(res[k]["host_ms"]), 3), first_call_ms=round(res[k]["first_call_ms"], 1))
    print(json.dumps(rec), flush=True)
m.engine.set_option("graph_split", 0)
