"""B=1 (M=1874) GEMM time vs K: T(K) = fixed + slope*K/64, per tile config, with and without the epilogue."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
import torch
E, lib, P, st, dev = mb.E, mb.lib, mb.P, mb.st, mb.dev
M = 1874
mb.gemm_case(M, 3072, 1024, 1, 1)
mb.gemm_case(M, 3072, 1024, 1, 1)
names = {1: "128x128", 2: "64x128reg", 3: "64x64reg", 5: "64x128ring", 6: "64x64ring"}
for N in (1024, 3072):
    for tile in (1, 2, 3, 5, 6):
        lib.f5_debug_set_gemm_tile(tile)
        for flags in (0, 1):
            lib.f5_debug_set_gemm_flags(flags)
            row = []
            for K in (64, 256, 512, 1024, 2048):
                a, w = mb.rnd(M, K), mb.rnd((N + 127) // 128 * 128, K)
                bias = torch.zeros(N, device=dev)
                oh = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
                fn = lambda: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(oh), P(None), M, N, K, K, K, N, 1, 1, st()))
                row.append(round(mb.timeit(fn, iters=50) * 1e3, 2))
            print(json.dumps(dict(N=N, tile=names[tile], skip_epi=flags, us_K64_256_512_1024_2048=row)), flush=True)
lib.f5_debug_set_gemm_flags(0)
lib.f5_debug_set_gemm_tile(0)
# skinny gemm timing (adaLN table: 31 x 137216 x 1024)
a = torch.randn(31, 1024, device=dev); w = torch.randn(137216, 1024, device=dev) * 0.03; b = torch.zeros(137216, device=dev)
out = torch.empty(31, 137216, device=dev)
fn = lambda: E.check(lib.f5_op_skinny_gemm(P(a), P(w), P(b), P(out), 31, 137216, 1024, 1, 0, st()))
ms = mb.timeit(fn, iters=10)
print(json.dumps(dict(op="skinny_gemm", ms=round(ms, 4), GBps=round(137216 * 1024 * 4 / ms / 1e6, 1))), flush=True)
