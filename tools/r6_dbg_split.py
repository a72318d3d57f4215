import sys, os
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, torch
from f5test import DEV, E, TINY, synth_inputs, synthetic_weights
from f5_tts_mlx_amd.cfm import F5TTS
from f5_tts_mlx_amd.dit import DiT
cfg = TINY
w = synthetic_weights(cfg, seed=42)
for prec in ("bf16", "f16"):
    m = DiT.from_config(cfg, precision=prec, device=DEV); m.load_weights(w)
    eng = m.engine; eng.split_batch = 0
    f5 = F5TTS(transformer=m)
    cond, text, durations, y0 = synth_inputs(cfg, 5, 96, nt=20, n_ref=16, seed=11, ragged=True)
    for g in (False, True):
        kw = dict(steps=5, y0=y0, method="euler", use_graph=g)
        eng.split_batch = 0
        full, tr = [x.clone() for x in f5.sample(cond, text, duration=torch.tensor(durations), **kw)]
        torch.cuda.synchronize()
        eng.split_batch = 4
        for rep in range(3):
            part, ptr_ = f5.sample(cond, text, duration=torch.tensor(durations), **kw)
            torch.cuda.synchronize()
            d = (part - full).abs()
            dt = (ptr_ - tr).abs()
            for i in range(5):
                rows = (d[i].amax(dim=1) > 0).nonzero().flatten().tolist()
                print(prec, "graph", g, "rep", rep, "utt", i, "dur", durations[i], "rows differing:", (rows[0], rows[-1], len(rows)) if rows else None, "max", float(d[i].max()),
                      "traj max", [float(dt[s, i].max()) for s in range(5)], "events", eng.split_events)
