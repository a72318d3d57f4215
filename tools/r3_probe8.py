"""Round-3 probe 8: interleaved A/B of the large-grid attention kernels at 64 x 16 heads x 937 (f16, q pre-multiplied):
wide 1 = f5_attn2f (4 waves x 64 queries, free-running), wide 2 = f5_attn2r (8 waves, role-split).  Rounds alternate so that
clock / power state drift shows up as round-to-round spread instead of as a kernel difference."""
import ctypes as C, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E
if os.environ.get("F5_PROBE_LIB"):                      # A/B against a library built from an earlier commit
    from pathlib import Path
    E._LIB_PATH = Path(os.environ["F5_PROBE_LIB"]).resolve()
from tools.r3_probe2 import ev_time, lib, dev, P, st
opd = torch.float16
H, D = 16, 1024
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2"])]
with E.operand_type("f16"):
    lib.f5_debug_set_op_q_premul(C.c_float(0.125 * 1.4426950408889634))
    for B, N in ((64, 937),):
        npad = (N + 63) // 64 * 64
        g = torch.Generator(device="cpu").manual_seed(B * N)
        qk = (torch.randn(B * N, 2 * D, generator=g) * 0.6).to(dev).to(opd)
        vt = torch.randn(B * H, 64, npad, generator=g).to(dev).to(opd)
        vt[:, :, N:] = 0
        ao = torch.zeros(B * N, D, dtype=opd, device=dev)
        fn = lambda: E.check(lib.f5_op_attention(P(qk), P(None), P(vt), P(None), P(ao), P(None), P(None), B, H, N, npad, D, C.c_float(0.125), 0, st()))
        def setw(w):
            # 3 = role-split kernel with the Q fragments in LDS (2 = in registers)
            E.check(lib.f5_debug_set_attn_wide(2 if w == 3 else w)); E.check(lib.f5_debug_set_attn_kvsplit(1 if w >= 0 else -1))
            if w >= 2: E.check(lib.f5_debug_set_attn_variant(32 if w == 3 else 0))     # (lab build only)
        setw(variants[0])
        for _ in range(200 if B > 8 else 2000): fn()
        torch.cuda.synchronize()
        res = {w: [] for w in variants}
        for r in range(5):
            for w in variants:
                setw(w)
                res[w].append(round(ev_time(fn, iters=30 if B > 8 else 200, warm=3), 1))
        print(json.dumps(dict(lib=os.path.basename(str(E._LIB_PATH)), B=B, N=N, us={str(k): v for k, v in res.items()},
                              tf={str(k): round(4.0 * B * H * N * N * 64 / min(v) / 1e6) for k, v in res.items()})), flush=True)
    E.check(lib.f5_debug_set_attn_wide(-1)); E.check(lib.f5_debug_set_attn_kvsplit(-1))
    lib.f5_debug_set_op_q_premul(C.c_float(0.0))
