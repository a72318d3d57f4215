"""Round 6: what the batch-1-sized LN fold costs and saves per launch (M = 2 x 937 rows, f16): the residual GEMMs (out-proj K = 1024, FF2
K = 2048) plain vs as fold producers (x16 + slice statistics), FF1 and QKV plain vs as
statistics-form consumers, and the LN-modulate launch the fold removes.  Back-to-back launches, HIP events."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402

lib = E.load_library()
dev = torch.device("cuda:0")
P = E.ptr
st = lambda: E.stream_ptr(dev)  # noqa: E731
B, N, D, FF, H = 2, 937, 1024, 2048, 16
M = B * N
npad = (N + 63) // 64 * 64


def ev(fn, iters=200, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


with E.operand_type("f16"):
    g = torch.Generator(device="cpu").manual_seed(0)
    rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc)  # noqa: E731
    a16 = rn(M, D).to(dev).half()
    a16f = rn(M, FF).to(dev).half()
    wo, w2 = rn(D, D, sc=D ** -0.5).to(dev).half(), rn(D, FF, sc=FF ** -0.5).to(dev).half()
    w1, wq = rn(FF, D, sc=D ** -0.5).to(dev).half(), rn(3 * D, D, sc=D ** -0.5).to(dev).half()
    bias, gate, sc, sh = rn(D).to(dev), rn(D).to(dev), rn(D, sc=0.3).to(dev), rn(D, sc=0.3).to(dev)
    bias1, biasq = rn(FF).to(dev), rn(3 * D).to(dev)
    x = rn(M, D).to(dev)
    x16 = torch.zeros(M, D, dtype=torch.float16, device=dev)
    stats = torch.zeros(16, M, 2, device=dev)
    shift = torch.zeros(M, device=dev)
    mean_out = torch.zeros(M, device=dev)
    h16 = torch.zeros(M, D, dtype=torch.float16, device=dev)
    out_ff = torch.zeros(M, FF, dtype=torch.float16, device=dev)
    c1 = torch.zeros(3 * D, device=dev)
    c2 = torch.zeros(3 * D, device=dev)
    cos_t, sin_t = torch.empty(N, 32, device=dev), torch.empty(N, 32, device=dev)
    E.check(lib.f5_op_rope_table(P(cos_t), P(sin_t), N, 64, st()))
    tt = [torch.empty(64 * N, device=dev) for _ in range(2)]
    E.check(lib.f5_op_rope_table_g4(P(tt[0]), P(tt[1]), N, 64, C.c_float(1.0), st()))
    qk = torch.zeros(M, 2 * D, dtype=torch.float16, device=dev)
    vt = torch.zeros(B * H, 64, npad, dtype=torch.float16, device=dev)
    res = {}

    def resid(a, w, K):
        return lambda: E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bias), P(gate), P(None), P(x), M, D, K, K, K, D, 1, st()))
    for nm, a, w, K in (("out_proj", a16, wo, D), ("ff2", a16f, w2, FF)):
        res[nm + "_plain"] = ev(resid(a, w, K))
        E.check(lib.f5_debug_set_op_fold_producer(P(sc), P(x16), P(stats), P(shift)))
        res[nm + "_producer"] = ev(resid(a, w, K))
        E.check(lib.f5_debug_set_op_fold_producer(P(None), P(None), P(None), P(None)))
    ff1 = lambda b: (lambda: E.check(lib.f5_op_gemm(P(x16), P(None), P(w1), P(None), P(b), P(None), P(out_ff), P(None), M, FF, D, D, D, FF, 1, 2, st())))  # noqa: E731
    res["ff1_plain"] = ev(ff1(bias1))
    E.check(lib.f5_debug_set_op_rope_tables_g4(P(tt[0]), P(tt[1])))
    qkv = lambda b: (lambda: E.check(lib.f5_op_qkv_rope(P(x16), P(None), P(wq), P(None), P(b), P(cos_t), P(sin_t), P(qk), P(None), P(vt), P(None),  # noqa: E731
                                                        B, N, npad, H, D, 1, st())))
    res["qkv_plain"] = ev(qkv(biasq))
    E.check(lib.f5_debug_set_op_fold_consumer(P(None), P(c1), P(c2)))
    E.check(lib.f5_debug_set_op_fold_stats(P(stats), M, P(shift), P(mean_out)))
    res["ff1_consumer_stats"] = ev(ff1(None))
    res["qkv_consumer_stats"] = ev(qkv(None))
    E.check(lib.f5_debug_set_op_fold_stats(P(None), 0, P(None), P(None)))
    E.check(lib.f5_debug_set_op_fold_consumer(P(None), P(None), P(None)))
    E.check(lib.f5_debug_set_op_rope_tables_g4(P(None), P(None)))
    res["ln_modulate"] = ev(lambda: E.check(lib.f5_op_ln_modulate(P(x), P(sc), P(sh), P(h16), P(None), M, D, st())))
    print(json.dumps({k: round(v, 2) for k, v in res.items()}))
