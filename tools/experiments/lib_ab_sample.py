"""Library-vs-library A/B at sample() level: run once per library in separate processes, alternating (F5TTS_HIP_LIB=<path to a libf5tts_hip.so>
python tools/experiments/lib_ab_sample.py); prints ms per sample() at batch 8 / 32 and a checksum of the output.  Round 4: the column vectors of the
residual epilogue (bias, gate, 1 + scale) requested before the K loop -- bit-identical, 1106.4 / 1107.1 vs 1106.9 / 1108.6 ms: no effect, not kept."""
import sys, os, time, json, torch
sys.path.insert(0, os.getcwd())
import bench
from f5_tts_mlx_amd.cfm import F5TTS
from f5_tts_mlx_amd.dit import DiT
from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights
dev = torch.device("cuda:0")
m = DiT.from_config(F5TTS_335M, precision="f16", device=dev)
m.load_weights(synthetic_weights(F5TTS_335M, seed=42))
f5 = F5TTS(transformer=m)
res = {}
for B in [int(b) for b in os.environ.get("F5_AB_BATCHES", "1,8,32").split(",")]:
    cond, text, y0, _ = bench.synth_batch(B, 0, dev)
    kw = dict(duration=bench.N_FRAMES, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, use_graph=True)
    ts = []
    for rep in range(int(os.environ.get("F5_AB_REPS", "3"))):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out, _ = f5.sample(cond, text, y0=y0, **kw)
        torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
    res[B] = [round(min(ts[1:]), 2), float(out.double().abs().sum())]
print(json.dumps(dict(lib=os.path.basename(os.environ.get("F5TTS_HIP_LIB", "libf5tts_hip.so")), res=res)))
