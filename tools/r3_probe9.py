"""Round-3 probe 9 (driven by r3_probe9.sh): loop one kernel for ~35 s so that rocm-smi can sample clock and power next to it."""
import ctypes as C, json, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E
from tools.r3_probe2 import ev_time, lib, dev, P, st
what = sys.argv[1]
opd = torch.float16
with E.operand_type("f16"):
    if what == "attn":
        lib.f5_debug_set_op_q_premul(C.c_float(0.125 * 1.4426950408889634))
        B, H, D, N = 64, 16, 1024, 937
        npad = (N + 63) // 64 * 64
        g = torch.Generator(device="cpu").manual_seed(1)
        qk = (torch.randn(B * N, 2 * D, generator=g) * 0.6).to(dev).to(opd)
        vt = torch.randn(B * H, 64, npad, generator=g).to(dev).to(opd)
        ao = torch.zeros(B * N, D, dtype=opd, device=dev)
        fn = lambda: E.check(lib.f5_op_attention(P(qk), P(None), P(vt), P(None), P(ao), P(None), P(None), B, H, N, npad, D, C.c_float(0.125), 0, st()))
        flop = 4.0 * B * H * N * N * 64
    else:
        M, N_, K = 59968, 2048, 1024     # FF1: 1024 -> 2048, GELU epilogue (epi 2)
        g = torch.Generator(device="cpu").manual_seed(2)
        a = (torch.randn(M, K, generator=g) * 0.5).to(dev).to(opd)
        w = (torch.randn(N_, K, generator=g) * 0.05).to(dev).to(opd)
        bias = torch.zeros(N_, dtype=torch.float32, device=dev)
        out = torch.zeros(M, N_, dtype=opd, device=dev)
        fn = lambda: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(out), P(None), M, N_, K, K, K, N_, 1, 2, st()))
        flop = 2.0 * M * N_ * K
    t_end = time.time() + 35
    us = []
    while time.time() < t_end:
        us.append(ev_time(fn, iters=200, warm=0))
    print(json.dumps(dict(what=what, first_us=round(us[0], 1), last_us=round(us[-1], 1), min_us=round(min(us), 1),
                          tf_last=round(flop / us[-1] / 1e6))))
