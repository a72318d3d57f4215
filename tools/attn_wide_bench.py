"""attention in-graph: 128-query workgroups (3 per CU) vs 256-query workgroups with two query blocks per wave."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
from tools.fixed_cost import graph_time
from tools.branch_split_lib import Chain, D, H, N, npad
import ctypes as C
E, lib, P = mb.E, mb.lib, mb.P
for nb in (2, 8, 64):
    c = Chain(nb)
    row = {}
    for wide in (0, 1):
        E.check(lib.f5_debug_set_attn_wide(wide)); E.check(lib.f5_debug_set_attn_kvsplit(1 if nb > 2 else -1))
        fn = lambda st: E.check(lib.f5_op_attention(P(c.qk), P(None), P(c.vt), P(None), P(c.ao), P(None), P(None), c.nb, H, N, npad, D, C.c_float(0.125), 0, st))
        us = graph_time(fn, reps=12 if nb > 8 else 44, iters=5)
        row[["v2", "wide", "pipelined"][wide]] = [round(us, 1), round(4.0 * nb * H * N * N * 64 / us / 1e6)]
    print(json.dumps(dict(nb=nb, us_tflops=row)), flush=True)
E.check(lib.f5_debug_set_attn_wide(-1)); E.check(lib.f5_debug_set_attn_kvsplit(-1))
