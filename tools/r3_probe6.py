"""Round-3 probe 6: is the large-grid attention kernel power (DVFS) limited like the 256x256 GEMM?  Same launch (64 x 16 heads x 937,
f16) on the workload's data, on zeros and on constant data; and the same at half the heads per launch."""
import ctypes as C, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E
from tools.r3_probe2 import ev_time, lib, dev, P, st
opd = torch.float16
B, H, N, D = 64, 16, 937, 1024
npad = 960
with E.operand_type("f16"):
    lib.f5_debug_set_op_q_premul(C.c_float(0.125 * 1.4426950408889634))
    fills = {
        "workload-like (q, k ~ N(0,1) scaled, v ~ N(0,1))": ((torch.randn(B * N, 2 * D, device=dev) * 0.6).to(opd), torch.randn(B * H, 64, npad, device=dev).to(opd)),
        "zeros": (torch.zeros(B * N, 2 * D, dtype=opd, device=dev), torch.zeros(B * H, 64, npad, dtype=opd, device=dev)),
        "constant 0.25": (torch.full((B * N, 2 * D), 0.25, dtype=opd, device=dev), torch.full((B * H, 64, npad), 0.25, dtype=opd, device=dev)),
    }
    ao = torch.empty(B * N, D, dtype=opd, device=dev)
    for name, (qk, vt) in fills.items():
        vt[:, :, N:] = 0
        fn = lambda: E.check(lib.f5_op_attention(P(qk), P(None), P(vt), P(None), P(ao), P(None), P(None), B, H, N, npad, D, C.c_float(0.125), 0, st()))
        us = ev_time(fn, iters=20, warm=5)
        print(json.dumps(dict(fill=name, us=round(us, 1), tf=round(4.0 * B * H * N * N * 64 / us / 1e6))), flush=True)
    lib.f5_debug_set_op_q_premul(C.c_float(0.0))
