import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
mb.gemm_case(1874, 1024, 1024, 0, 1)
for order in (1, 2, 1, 2):
    mb.lib.f5_debug_set_gemm_order(order)
    print("order", order)
    for (N, K, epi) in ((3072, 1024, 1), (1024, 1024, 0), (2048, 1024, 2), (1024, 2048, 0)):
        mb.gemm_case(1874, N, K, epi, 1)
mb.lib.f5_debug_set_gemm_order(0)
