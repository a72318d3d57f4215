"""Round-4 probe: per-workgroup fixed cost vs per-tile cost of the large-grid attention kernels (time / rounds = a + b * tiles)."""
import ctypes as C, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E
from tools.yardstick import ev_time, lib, dev, P, st
opd = torch.float16
H, D = 16, 1024
names = {0: "v2f", 1: "v2p"}
with E.operand_type("f16"):
    lib.f5_debug_set_op_q_premul(C.c_float(0.125 * 1.4426950408889634))
    E.check(lib.f5_debug_set_attn_wide(1)); E.check(lib.f5_debug_set_attn_kvsplit(1))
    for B, N in ((64, 937), (64, 425), (32, 1960), (16, 3008)):
        npad = (N + 63) // 64 * 64
        g = torch.Generator(device="cpu").manual_seed(N)
        qk = (torch.randn(B * N, 2 * D, generator=g) * 0.6).to(dev).to(opd)
        vt = torch.randn(B * H, 64, npad, generator=g).to(dev).to(opd)
        vt[:, :, N:] = 0
        ao = torch.zeros(B * N, D, dtype=opd, device=dev)
        fn = lambda: E.check(lib.f5_op_attention(P(qk), P(None), P(vt), P(None), P(ao), P(None), P(None), B, H, N, npad, D, C.c_float(0.125), 0, st()))
        wgs = B * H * ((N + 255) // 256)
        tiles = (N + 63) // 64
        rec = dict(B=B, N=N, wgs=wgs, rounds=wgs / 256, tiles_per_wg=tiles, us={}, us_per_round={}, tflops={})
        for rnd in range(2):
            for v in names:
                E.check(lib.f5_debug_set_attn_pipe(v))
                rec["us"].setdefault(names[v], []).append(round(ev_time(fn, iters=10), 1))
        for k, v in rec["us"].items():
            rec["us_per_round"][k] = round(min(v) / (wgs / 256), 2)
            rec["tflops"][k] = round(4.0 * B * H * N * N * 64 / min(v) / 1e6)
        print(json.dumps(rec), flush=True)
    E.check(lib.f5_debug_set_attn_pipe(0)); E.check(lib.f5_debug_set_attn_wide(-1)); E.check(lib.f5_debug_set_attn_kvsplit(-1))
    lib.f5_debug_set_op_q_premul(C.c_float(0.0))
