import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
import ctypes as C, torch, json
E, lib, P, st, dev = mb.E, mb.lib, mb.P, mb.st, mb.dev
def resid_case(M, N, K):
    a, w = mb.rnd(M, K), mb.rnd(N, K)
    bias, gate = torch.zeros(N, device=dev), torch.ones(N, device=dev)
    x = torch.zeros(M, N, device=dev)
    fn = lambda: E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bias), P(gate), P(None), P(x), M, N, K, K, K, N, 1, st()))
    ms = mb.timeit(fn)
    print(json.dumps(dict(op="gemm_resid", M=M, N=N, K=K, ms=round(ms, 4), tflops=round(2.0 * M * N * K / ms / 1e9, 1))), flush=True)
mb.gemm_case(59968, 1024, 1024, 0, 1)   # warm-up
import itertools
for big, stagger, flags in ((2, -1, 0), (3, -1, 0), (3, 0, 0), (3, 30000, 0), (3, -1, 1), (2, -1, 0), (3, -1, 0)):
    lib.f5_debug_set_gemm_big_kernel(big, stagger)
    lib.f5_debug_set_gemm_flags(flags)
    print("big", big, "stagger", stagger, "flags", flags)
    mb.gemm_case(59968, 3072, 1024, 1, 1)
    mb.gemm_case(59968, 2048, 1024, 2, 1)
    mb.gemm_case(59968, 1024, 1024, 0, 1)
    resid_case(59968, 1024, 1024)
    resid_case(59968, 1024, 2048)
lib.f5_debug_set_gemm_flags(0)
lib.f5_debug_set_gemm_big_kernel(2, -1)
