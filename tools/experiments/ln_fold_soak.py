"""Repeatability screen of the folded LN path: batch-32 sample() (8-point Euler), 12 repeats alternating eager / graph replay, outputs
compared bitwise with the first (round 4: 0 mismatches).  python tools/experiments/ln_fold_soak.py"""
import sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
from f5_tts_mlx_amd.cfm import F5TTS
from f5_tts_mlx_amd.dit import DiT
from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights
dev = torch.device("cuda:0")
m = DiT.from_config(F5TTS_335M, precision="f16", device=dev)
m.load_weights(synthetic_weights(F5TTS_335M, seed=42))
f5 = F5TTS(transformer=m)
cond, text, y0, _ = bench.synth_batch(32, 0, dev)
kw = dict(duration=bench.N_FRAMES, steps=8, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0)
ref = None
bad = 0
for rep in range(12):
    out, _ = f5.sample(cond, text, y0=y0, use_graph=(rep % 2 == 1), **kw)
    torch.cuda.synchronize()
    if ref is None: ref = out.clone()
    elif not torch.equal(out, ref): bad += 1
print("ln_fold option", m.engine.get_option("ln_fold"), "repeats 12 (eager / graph alternating), mismatches vs first:", bad, "finite", bool(torch.isfinite(ref).all()))
