"""256x256 bf16 GEMM at the B=32 shapes of a DiT block (in-graph), with and without the epilogue."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
from tools.fixed_cost import graph_time
import ctypes as C, torch
E, lib, P, dev = mb.E, mb.lib, mb.P, mb.dev
M = 59968
for (N, K, epi, name) in ((3072, 1024, 1, "qkv-shape bf16 out"), (2048, 1024, 2, "ff1 gelu"), (1024, 1024, 4, "oproj resid"), (1024, 2048, 4, "ff2 resid")):
    a, w = mb.rnd(M, K), mb.rnd(N, K)
    bias = torch.zeros(N, device=dev); gate = torch.ones(N, device=dev) * 0.01
    of = torch.zeros(M, N, device=dev); ob = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    if epi == 4:
        fn = lambda st: E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bias), P(gate), P(None), P(of), M, N, K, K, K, N, 1, st))
    else:
        fn = lambda st: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(ob), P(None), M, N, K, K, K, N, 1, epi, st))
    row = {}
    for flags in (1, 0, 1, 0):
        E.check(lib.f5_debug_set_gemm_flags(flags))
        us = graph_time(fn, reps=8, iters=5)
        row.setdefault("noepi" if flags else "full", []).append([round(us, 1), round(2.0 * M * N * K / us / 1e6)])
    E.check(lib.f5_debug_set_gemm_flags(0))
    print(json.dumps(dict(op=name, us_tflops=row)), flush=True)
