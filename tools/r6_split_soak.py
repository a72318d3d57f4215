"""Round 6: soak of the split sample() path (Engine.split_batch): random batch sizes, lengths, solvers and graph modes on the tiny model,
every result compared bitwise with the unsplit call.  Shapes outnumber the graph cache (8 entries), so graphs are evicted and captured
again -- from two host threads at once.  usage: python tools/r6_split_soak.py [iterations]"""
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5test import DEV, TINY, synth_inputs, synthetic_weights  # noqa: E402
from f5_tts_mlx_amd.cfm import F5TTS  # noqa: E402
from f5_tts_mlx_amd.dit import DiT  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 300
r = np.random.default_rng(123)
w = synthetic_weights(TINY, seed=42)
bad, t0 = [], time.time()
for prec in ("f16", "bf16"):
    m = DiT.from_config(TINY, precision=prec, device=DEV)
    m.load_weights(w)
    f5 = F5TTS(transformer=m)
    eng = m.engine
    shapes = [(int(r.integers(2, 8)), int(r.integers(40, 130)), bool(r.integers(0, 2))) for _ in range(14)]
    for it in range(iters):
        B, N, ragged = shapes[int(r.integers(0, len(shapes)))]
        method, steps = [("euler", 5), ("midpoint", 4), ("rk4", 3)][int(r.integers(0, 3))]
        g = [True, False, "auto"][int(r.integers(0, 3))]
        cfg = [2.0, 0.0, 0.7][int(r.integers(0, 3))]
        cond, text, durations, y0 = synth_inputs(TINY, B, N, nt=16, n_ref=12, seed=it % 7, ragged=ragged)
        kw = dict(duration=torch.tensor(durations), steps=steps, method=method, y0=y0, cfg_strength=cfg)
        eng.split_batch = 0
        want = [x.clone() for x in f5.sample(cond, text, use_graph=False, **kw)]
        eng.split_batch = 2
        got = f5.sample(cond, text, use_graph=g, **kw)
        torch.cuda.synchronize()
        if not (torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])):
            bad.append((prec, it, B, N, ragged, method, str(g), cfg))
    print(json.dumps(dict(probe="split_soak", precision=prec, iterations=iters, split_calls=eng.split_events, graphs_cached=eng.graph_count(),
                          sibling_graphs=eng._sibling.graph_count(), mismatches=bad, seconds=round(time.time() - t0, 1))), flush=True)
sys.exit(1 if bad else 0)
