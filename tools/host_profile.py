"""Host-side cost of F5TTS.sample() at B=1 (wall - GPU time): cProfile of 20 calls."""
import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as BN
from f5_tts_mlx_amd.cfm import F5TTS
from f5_tts_mlx_amd.dit import DiT
from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights
dev = torch.device("cuda:0")
model = DiT.from_config(F5TTS_335M, precision="bf16", device=dev)
model.load_weights(synthetic_weights(F5TTS_335M, seed=42))
f5 = F5TTS(transformer=model)
cond, text, y0 = BN.synth_batch(1, 0, dev)
kw = dict(duration=BN.N_FRAMES, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0, use_graph=True)
for _ in range(3):
    f5.sample(cond, text, **kw)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0 = time.perf_counter(); e0.record()
for _ in range(10):
    out, _ = f5.sample(cond, text, **kw)
e1.record(); torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / 10 * 1e3
print(f"wall {wall:.2f} ms per sample, GPU-event span {e0.elapsed_time(e1) / 10:.2f} ms")
# time until sample() returns (host) vs until the GPU is done
t0 = time.perf_counter()
out, _ = f5.sample(cond, text, **kw)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"sample() returns after {1e3 * (t1 - t0):.2f} ms, GPU done after {1e3 * (t2 - t0):.2f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(10):
    out, _ = f5.sample(cond, text, **kw)
torch.cuda.synchronize(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:4000])
