"""Replica of the call pattern that produced garbage (r2f variant N): 4 synchronised samples, then back-to-back loops."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B
from f5_tts_mlx_amd.cfm import F5TTS
from f5_tts_mlx_amd.dit import DiT
from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights
from f5_tts_mlx_amd import engine as E
dev = torch.device("cuda:0")
g = np.load(B.GOLDEN)
def par(t): return float(np.abs(t.detach().cpu().numpy().astype(np.float64) - g["out"]).mean())
def csum(t): return int(t.contiguous().view(torch.uint8).to(torch.int64).sum().item())
variant = sys.argv[1] if len(sys.argv) > 1 else "N"
prec = sys.argv[2] if len(sys.argv) > 2 else "f16"
model = DiT.from_config(F5TTS_335M, precision=prec, device=dev)
model.load_weights(synthetic_weights(F5TTS_335M, seed=42))
keepalive = []
f5 = F5TTS(transformer=model)
cond, text, y0, waves = B.synth_batch(1, 0, dev)
kw = dict(duration=B.N_FRAMES, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0)
for mode in (False, True, True, True):
    out, _ = f5.sample(cond, text, use_graph=mode, **kw)
    torch.cuda.synchronize()
    print(f"{variant} sample use_graph={mode}: parity {par(out[0]):.3e}")
eng = model.engine
s0 = dict(arena=csum(eng.arena), y0=csum(y0), text=csum(text), cond=csum(cond))
res = []
out = None
for i in range(7):
    out, _ = f5.sample(cond, text, use_graph=True, **kw)
    res.append(out.clone())
torch.cuda.synchronize()
print(variant, "loop, clone per call:", " ".join(f"{par(o[0]):.2e}" for o in res), "| final out:", f"{par(out[0]):.2e}")
s1 = dict(arena=csum(eng.arena), y0=csum(y0), text=csum(text), cond=csum(cond))
print(variant, "checksums changed:", {k: (s0[k], s1[k]) for k in s0 if s0[k] != s1[k]})
out, tr = f5.sample(cond, text, use_graph=True, **kw)
torch.cuda.synchronize()
print(variant, "synchronised graph sample afterwards:", f"{par(out[0]):.2e}", "traj[1] absmax", float(tr[1].abs().max()), "traj[0]==y0", bool(torch.equal(tr[0], y0)))
out, tr = f5.sample(cond, text, use_graph=False, **kw)
torch.cuda.synchronize()
print(variant, "synchronised eager sample afterwards:", f"{par(out[0]):.2e}")
