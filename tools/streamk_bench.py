"""B=32 (M=59968) GEMMs of one DiT block: stream-K schedule vs one tile per workgroup (256x256 kernel)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
from tools.fixed_cost import graph_time
from tools.branch_split_lib import Chain, D, FF, H, N, npad
import ctypes as C, torch
E, lib, P, dev = mb.E, mb.lib, mb.P, mb.dev
for nb in (64,):
    c = Chain(nb)
    cos_t = torch.empty((N, 32), device=dev); sin_t = torch.empty((N, 32), device=dev)
    E.check(lib.f5_op_rope_table(P(cos_t), P(sin_t), N, 64, E.stream_ptr(dev)))
    qkb = mb.rnd(c.M, 2 * D)
    ops = dict(
        qkv=(lambda st: E.check(lib.f5_op_qkv_rope(P(c.h), P(None), P(c.wqkv), P(None), P(c.bq), P(cos_t), P(sin_t), P(qkb), P(None), P(c.vt), P(None), c.nb, N, npad, H, D, 1, st)), 2.0 * c.M * 3 * D * D),
        ff1=(lambda st: E.check(lib.f5_op_gemm(P(c.h), P(None), P(c.w1), P(None), P(c.b1), P(None), P(c.ff), P(None), c.M, FF, D, D, D, FF, 1, 2, st)), 2.0 * c.M * FF * D),
        oproj=(lambda st: E.check(lib.f5_op_gemm_resid_gate(P(c.ao), P(None), P(c.wo), P(None), P(c.bo), P(c.gate), P(None), P(c.x), c.M, D, D, D, D, D, 1, st)), 2.0 * c.M * D * D),
        ff2=(lambda st: E.check(lib.f5_op_gemm_resid_gate(P(c.ff), P(None), P(c.w2), P(None), P(c.bo), P(c.gate), P(None), P(c.x), c.M, D, FF, FF, FF, D, 1, st)), 2.0 * c.M * D * FF),
    )
    for k, (fn, flops) in ops.items():
        row = {}
        for sk in (2, 0):
            for flags in (0, 1):
                E.check(lib.f5_debug_set_gemm_streamk(sk)); E.check(lib.f5_debug_set_gemm_flags(flags))
                us = graph_time(fn, reps=8, iters=5)
                row[f"sk{sk}{'_noepi' if flags else ''}"] = [round(us, 1), round(flops / us / 1e6, 0)]
        E.check(lib.f5_debug_set_gemm_flags(0)); E.check(lib.f5_debug_set_gemm_streamk(0))
        print(json.dumps(dict(nb=nb, op=k, us_tflops=row)), flush=True)
print("streamk_error", lib.f5_debug_gemm_streamk_error())
