"""B=1 (M=1874) QKV / FF1 / out-proj / FF2 GEMMs in-graph: current auto choice vs the 8-wave ring tiles."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
from tools.fixed_cost import graph_time
from tools.branch_split_lib import Chain, D, FF, H, N, npad
import ctypes as C, torch
E, lib, P, dev = mb.E, mb.lib, mb.P, mb.dev
for nb in (2, 1):
    c = Chain(nb)
    cos_t = torch.empty((N, 32), device=dev); sin_t = torch.empty((N, 32), device=dev)
    E.check(lib.f5_op_rope_table(P(cos_t), P(sin_t), N, 64, E.stream_ptr(dev)))
    qkb = mb.rnd(c.M, 2 * D)
    ops = dict(
        qkv=lambda st: E.check(lib.f5_op_qkv_rope(P(c.h), P(None), P(c.wqkv), P(None), P(c.bq), P(cos_t), P(sin_t), P(qkb), P(None), P(c.vt), P(None), c.nb, N, npad, H, D, 1, st)),
        ff1=lambda st: E.check(lib.f5_op_gemm(P(c.h), P(None), P(c.w1), P(None), P(c.b1), P(None), P(c.ff), P(None), c.M, FF, D, D, D, FF, 1, 2, st)),
        oproj=lambda st: E.check(lib.f5_op_gemm_resid_gate(P(c.ao), P(None), P(c.wo), P(None), P(c.bo), P(c.gate), P(None), P(c.x), c.M, D, D, D, D, D, 1, st)),
        ff2=lambda st: E.check(lib.f5_op_gemm_resid_gate(P(c.ff), P(None), P(c.w2), P(None), P(c.bo), P(c.gate), P(None), P(c.x), c.M, D, FF, FF, FF, D, 1, st)),
    )
    for k, fn in ops.items():
        row = {}
        for tile in (0, 9, 10, 11):
            for flags in (0,):
                E.check(lib.f5_debug_set_gemm_tile(tile))
                E.check(lib.f5_debug_set_gemm_flags(flags))
                row[f"{tile}{'d' if flags else ''}"] = round(graph_time(fn), 2)
        E.check(lib.f5_debug_set_gemm_flags(0))
        print(json.dumps(dict(nb=nb, op=k, us_by_tile=row)), flush=True)
E.check(lib.f5_debug_set_gemm_tile(0))
