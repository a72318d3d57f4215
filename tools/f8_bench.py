"""MX-fp8 vs bf16 256x256 GEMM at the B=32 shapes of a DiT block (in-graph)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
from tools.fixed_cost import graph_time
import ctypes as C, torch
E, lib, P, dev = mb.E, mb.lib, mb.P, mb.dev
M = 59968
for (N, K, epi, name) in ((3072, 1024, 1, "qkv-shape bf16 out"), (2048, 1024, 2, "ff1 gelu"), (1024, 1024, 4, "oproj resid"), (1024, 2048, 4, "ff2 resid")):
    a, w = mb.rnd(M, K), mb.rnd(N, K)
    a8 = torch.randint(0, 120, (M, K), dtype=torch.uint8, device=dev); w8 = torch.randint(0, 120, (N, K), dtype=torch.uint8, device=dev)
    asc = torch.full((M, K // 32), 120, dtype=torch.uint8, device=dev); wsc = torch.full((N, K // 32), 120, dtype=torch.uint8, device=dev)
    bias = torch.zeros(N, device=dev); gate = torch.ones(N, device=dev) * 0.01
    of = torch.zeros(M, N, device=dev); ob = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    o8 = torch.empty(M, N, dtype=torch.uint8, device=dev); o8s = torch.empty(M, N // 32, dtype=torch.uint8, device=dev)
    row = {}
    if epi == 4:
        fb = lambda st: E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bias), P(gate), P(None), P(of), M, N, K, K, K, N, 1, st))
    else:
        fb = lambda st: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(ob), P(None), M, N, K, K, K, N, 1, epi, st))
    f8 = lambda st: E.check(lib.f5_op_gemm_f8(P(a8), P(asc), P(w8), P(wsc), P(bias), P(gate), P(None), P(of), P(ob), P(o8), P(o8s), M, N, K, K, K, N, epi, st))
    for nm, fn in (("bf16", fb), ("mxfp8", f8)):
        us = graph_time(fn, reps=8, iters=5)
        row[nm] = [round(us, 1), round(2.0 * M * N * K / us / 1e6)]
    print(json.dumps(dict(op=name, N=N, K=K, us_tflops=row)), flush=True)
