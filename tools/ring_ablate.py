"""What a K step of a one-round ring GEMM is made of (batch-1 shapes, in-graph timing, garbage results): tile 13 (128x256, 8 waves)
at the QKV shape and tile 10 (64x128, two K groups of 4 waves) at the out-proj / FF2 shapes, with parts of the main loop removed."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
from tools.fixed_cost import graph_time
import torch
E, lib, P, dev = mb.E, mb.lib, mb.P, mb.dev
NAMES = {0: "full", 16: "no_loads", 32: "no_mfma", 64: "no_ldsread", 128: "no_barrier", 96: "loads+barrier", 112: "barrier_only",
         144: "ldsread+mfma", 240: "skeleton"}
for tile, M, N, K in ((13, 1874, 3072, 1024), (13, 937, 3072, 1024), (10, 1874, 1024, 1024), (10, 1874, 1024, 2048)):
    a, w = mb.rnd(M, K), mb.rnd(N, K)
    bias = torch.zeros(N, device=dev)
    ob = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fn = lambda st: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(ob), P(None), M, N, K, K, K, N, 1, 1, st))
    E.check(lib.f5_debug_set_gemm_tile(tile))
    row = {}
    for flags, name in NAMES.items():
        E.check(lib.f5_debug_set_gemm_flags(flags))
        row[name] = round(graph_time(fn), 2)
    E.check(lib.f5_debug_set_gemm_flags(0))
    E.check(lib.f5_debug_set_gemm_tile(0))
    print(json.dumps(dict(tile=tile, M=M, N=N, K=K, ktiles=K // 64, us=row)), flush=True)
# the same QKV-shaped GEMM with the plain bf16 epilogue on every tile shape (full main loops)
M, N, K = 1874, 3072, 1024
a, w = mb.rnd(M, K), mb.rnd(N, K)
bias = torch.zeros(N, device=dev)
ob = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
fn = lambda st: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(ob), P(None), M, N, K, K, K, N, 1, 1, st))
row = {}
for rep in range(2):
    for tile in (0, 2, 5, 8, 9, 12, 13):
        E.check(lib.f5_debug_set_gemm_tile(tile))
        row.setdefault(str(tile), []).append(round(graph_time(fn), 2))
E.check(lib.f5_debug_set_gemm_tile(0))
print(json.dumps(dict(op="qkv-shape bf16 out", M=M, N=N, K=K, us_by_tile=row)), flush=True)
