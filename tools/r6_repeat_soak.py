"""Round 6: repeatability soak of the DEFAULT path on the final library: the same sample() call repeated (batch 1: 200 x graph replay + 20
eager; batch 32: 16 x graph replay + 2 eager; 32-point Euler, 335M), every result bitwise equal to the first, status word clean."""
import json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from f5_tts_mlx_amd.cfm import F5TTS
from f5_tts_mlx_amd.dit import DiT
from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights
dev = torch.device("cuda:0")
m = DiT.from_config(F5TTS_335M, precision="f16", device=dev)
m.load_weights(synthetic_weights(F5TTS_335M, seed=42))
f5 = F5TTS(transformer=m)
bad = 0
for B, n_graph, n_eager in ((1, 200, 20), (32, 16, 2)):
    cond, text, y0, _ = bench.synth_batch(B, 0, dev)
    kw = dict(duration=bench.N_FRAMES, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
    ref = f5.sample(cond, text, use_graph=True, **kw)[0].clone()
    t0 = time.time()
    mism = 0
    for i in range(n_graph + n_eager):
        out = f5.sample(cond, text, use_graph=i < n_graph, **kw)[0]
        torch.cuda.synchronize()
        mism += int(not torch.equal(out, ref))
    bad += mism
    print(json.dumps(dict(probe="repeat_soak", batch=B, graph_calls=n_graph, eager_calls=n_eager, mismatches=mism, finite=bool(torch.isfinite(ref).all()),
                          range_events=m.engine.range_events, saturation_events=m.engine.saturation_events, seconds=round(time.time() - t0, 1))), flush=True)
sys.exit(1 if bad else 0)
