"""in-graph per-launch floor: GEMM (M=1874, N=1024) vs K, LN, and an almost empty kernel chain."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
from tools.fixed_cost import graph_time
import ctypes as C, torch
E, lib, P, dev = mb.E, mb.lib, mb.P, mb.dev
M, N = 1874, 1024
for tile in (6, 10, 3):
    E.check(lib.f5_debug_set_gemm_tile(tile))
    row = []
    for K in (64, 128, 256, 512, 1024, 2048):
        a, w = mb.rnd(M, K), mb.rnd(N, K)
        bias = torch.zeros(N, device=dev); oh = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
        fn = lambda st: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(oh), P(None), M, N, K, K, K, N, 1, 1, st))
        row.append(round(graph_time(fn), 2))
    print(json.dumps(dict(tile=tile, us_K64_128_256_512_1024_2048=row)), flush=True)
E.check(lib.f5_debug_set_gemm_tile(0))
t = torch.zeros(8, device=dev); s_ = torch.empty(8, 256, device=dev)
fn = lambda st: E.check(lib.f5_op_time_sinus(P(t), P(s_), 8, 256, st))
print(json.dumps(dict(op="tiny kernel (time_sinus, 8 rows)", us=round(graph_time(fn), 2))))
x = torch.zeros(1874, 1024, device=dev); sc = torch.zeros(1024, device=dev); h = torch.empty(1874, 1024, dtype=torch.bfloat16, device=dev)
fn = lambda st: E.check(lib.f5_op_ln_modulate(P(x), P(sc), P(sc), P(h), P(None), 1874, 1024, st))
print(json.dumps(dict(op="ln_modulate 1874x1024", us=round(graph_time(fn), 2))))
