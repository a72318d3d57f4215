"""v2 attention ablations in-graph at nb=2 (B=1 CFG) and nb=8: which phase carries the per-tile time?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
from tools.fixed_cost import graph_time
from tools.branch_split_lib import Chain, D, H, N, npad
import ctypes as C, torch
E, lib, P, dev = mb.E, mb.lib, mb.P, mb.dev
E.check(lib.f5_debug_set_attn_kvsplit(1))
names = {0: "full", 1: "no_exp2", 2: "no_barrier_wait", 3: "no_PV_mfma", 4: "no_S_mfma", 5: "no_softmax_valu", 6: "no_kv_loads", 7: "no_lds_reads"}
for nb in (2, 8, 32):
    c = Chain(nb)
    row = {}
    for abl in (0, 2, 5, 6, 7):
        E.check(lib.f5_debug_set_attn_ablation(abl))
        fn = lambda st: E.check(lib.f5_op_attention(P(c.qk), P(None), P(c.vt), P(None), P(c.ao), P(None), P(None), c.nb, H, N, npad, D, C.c_float(0.125), 0, st))
        row[names[abl]] = round(graph_time(fn, reps=22 if nb > 8 else 44), 2)
    print(json.dumps(dict(nb=nb, us=row)), flush=True)
E.check(lib.f5_debug_set_attn_ablation(0))
E.check(lib.f5_debug_set_attn_kvsplit(-1))
