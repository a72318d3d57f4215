"""QKV projection (+ RoPE, head split) at the batch-1 shapes (M = 1874 / 937, N = 3072, K = 1024), in-graph: the auto choice
(register-staged 64x128 tiles, 3 workgroups per CU) vs the ring tiles 8 (128x192) / 9 (128x128) and the 128x256 ring tiles
12 / 13 (8 waves of 64x64, 8 waves of 32x128)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
from tools.fixed_cost import graph_time
from tools.branch_split_lib import Chain, D, FF, H, N, npad
import torch
E, lib, P, dev = mb.E, mb.lib, mb.P, mb.dev
for nb in (2, 1):
    c = Chain(nb)
    cos_t = torch.empty((N, 32), device=dev); sin_t = torch.empty((N, 32), device=dev)
    E.check(lib.f5_op_rope_table(P(cos_t), P(sin_t), N, 64, E.stream_ptr(dev)))
    qkb = mb.rnd(c.M, 2 * D)
    fn = lambda st: E.check(lib.f5_op_qkv_rope(P(c.h), P(None), P(c.wqkv), P(None), P(c.bq), P(cos_t), P(sin_t), P(qkb), P(None), P(c.vt), P(None), c.nb, N, npad, H, D, 1, st))
    row = {}
    for rep in range(2):
        for tile in (0, 8, 12, 13):
            E.check(lib.f5_debug_set_gemm_tile(tile))
            row.setdefault(str(tile), []).append(round(graph_time(fn), 2))
    E.check(lib.f5_debug_set_gemm_tile(0))
    print(json.dumps(dict(nb=nb, M=c.M, op="qkv", us_by_tile=row)), flush=True)
