"""Does polling rocm-smi (bench.py ClockPowerSampler) slow the batch-32 calls it watches?  Alternating legs with / without, same process."""
import json, os, sys, time, statistics
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
cond, text, y0, _ = bench.synth_batch(32, 0, dev)
kw = dict(duration=bench.N_FRAMES, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
f5.sample(cond, text, **kw); torch.cuda.synchronize()
def leg(n=4):
    out = []
    for _ in range(n):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f5.sample(cond, text, **kw); torch.cuda.synchronize()
        out.append(round((time.perf_counter() - t0) * 1e3, 1))
    return out
rec = dict(probe="sampler_cost", legs=[])
for rnd in range(3):
    rec["legs"].append({"plain": leg()})
    with bench.ClockPowerSampler() as s:
        rec["legs"].append({"sampled": leg(), "smi": None})
    rec["legs"][-1]["smi"] = s.summary()
print(json.dumps(rec))
