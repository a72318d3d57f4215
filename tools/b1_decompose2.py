import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
import torch
E, lib, P, st, dev = mb.E, mb.lib, mb.P, mb.st, mb.dev
M = 1874
mb.gemm_case(M, 3072, 1024, 1, 1)
names = {1: "128x128", 4: "256x256v2", 7: "128x256v3"}
for N in (1024, 2048, 3072):
    for tile in (1, 4, 7):
        lib.f5_debug_set_gemm_tile(tile)
        row = []
        for K in (64, 256, 512, 1024, 2048):
            a, w = mb.rnd(M, K), mb.rnd((N + 255) // 256 * 256, K)
            bias = torch.zeros(N, device=dev)
            oh = torch.empty(M, N, dtype=torch.bfloat16, device=dev)
            fn = lambda: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(oh), P(None), M, N, K, K, K, N, 1, 1, st()))
            row.append(round(mb.timeit(fn, iters=50) * 1e3, 2))
        print(json.dumps(dict(N=N, tile=names[tile], us_K64_256_512_1024_2048=row)), flush=True)
lib.f5_debug_set_gemm_tile(0)
