import sys, os, json, time, numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from f5_tts_mlx_amd.cfm import F5TTS
from f5_tts_mlx_amd.dit import DiT
from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights
dev = torch.device("cuda:0")
m = DiT.from_config(F5TTS_335M, precision="f16", device=dev)
m.load_weights(synthetic_weights(F5TTS_335M, seed=42))
f5 = F5TTS(transformer=m)
cond, text, y0, _ = bench.synth_batch(1, 0, dev)
kw = dict(duration=bench.N_FRAMES, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
g = np.load("tests/golden/full_b1_euler32.npz")["out"]
res = {}
for fold in (0, -1):
    m.engine.set_option("ln_fold", fold)
    out, _ = f5.sample(cond, text, use_graph=False, **kw)
    torch.cuda.synchronize()
    out2, _ = f5.sample(cond, text, use_graph=True, **kw)
    torch.cuda.synchronize()
    res[fold] = out.cpu()
    print("ln_fold", fold, "L1 vs oracle", float(np.abs(out.cpu().numpy()[0].astype(np.float64) - g).mean()), "graph==eager", torch.equal(out, out2), "events", m.engine.range_events, m.engine.saturation_events, flush=True)
print("folded vs unfolded", float((res[0] - res[-1]).abs().mean()), "identical", torch.equal(res[0], res[-1]))
