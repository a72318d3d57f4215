"""Is a persistent one-workgroup-per-CU launch of the 256x256 kernel slower even without any split tile?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
from tools.fixed_cost import graph_time
import ctypes as C, torch
E, lib, P, dev = mb.E, mb.lib, mb.P, mb.dev
for (M, N, K) in ((65536, 2048, 1024), (65536, 1024, 1024), (16384, 4096, 1024)):
    a, w = mb.rnd(M, K), mb.rnd(N, K)
    bias = torch.zeros(N, device=dev); oh = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fn = lambda st: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(oh), P(None), M, N, K, K, K, N, 1, 1, st))
    row = {}
    for sk in (0, 2, 1):
        for flags in (1, 0):
            E.check(lib.f5_debug_set_gemm_streamk(sk)); E.check(lib.f5_debug_set_gemm_flags(flags))
            us = graph_time(fn, reps=6, iters=4)
            row[f"sk{sk}{'_noepi' if flags else ''}"] = [round(us, 1), round(2.0 * M * N * K / us / 1e6)]
    E.check(lib.f5_debug_set_gemm_flags(0)); E.check(lib.f5_debug_set_gemm_streamk(0))
    print(json.dumps(dict(M=M, N=N, K=K, tiles=(M // 256) * (N // 256), us_tflops=row)), flush=True)
