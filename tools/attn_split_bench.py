"""attention kernel time in-graph vs KV split factor and vs key length (fixed cost vs per-tile cost) at small batch."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
from tools.fixed_cost import graph_time
from tools.branch_split_lib import Chain, D, H, N, npad
import ctypes as C, torch
E, lib, P, dev = mb.E, mb.lib, mb.P, mb.dev
nb = 2
c = Chain(nb)
for ks in (1, 2, 4):
    E.check(lib.f5_debug_set_attn_kvsplit(ks))
    row = []
    for kv in (64, 256, 512, 937):
        kvl = torch.full((nb,), kv, dtype=torch.int32, device=dev)
        fn = lambda st: E.check(lib.f5_op_attention(P(c.qk), P(None), P(c.vt), P(None), P(c.ao), P(None), P(kvl), c.nb, H, N, npad, D, C.c_float(0.125), 0, st))
        row.append(round(graph_time(fn), 2))
    print(json.dumps(dict(nb=nb, ks=ks, us_kv64_256_512_937=row)), flush=True)
E.check(lib.f5_debug_set_attn_kvsplit(-1))
