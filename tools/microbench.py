"""Kernel micro-benchmarks on the GPU box (HIP events via torch on the launch stream). Prints one line per case."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402

lib = E.load_library()
dev = torch.device("cuda:0")
st = lambda: E.stream_ptr(dev)
P = E.ptr


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def rnd(*s):
    return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)


def gemm_case(M, N, K, epi, nseg):
    a, al, w, wl = rnd(M, K), rnd(M, K), rnd((N + 127) // 128 * 128, K), rnd((N + 127) // 128 * 128, K)
    bias = torch.zeros(N, device=dev)
    of = torch.empty(M, N, device=dev)
    oh, ol = torch.empty(M, N, dtype=torch.bfloat16, device=dev), torch.empty(M, N, dtype=torch.bfloat16, device=dev)
    fn = lambda: E.check(lib.f5_op_gemm(P(a), P(al), P(w), P(wl), P(bias), P(of), P(oh), P(ol), M, N, K, K, K, N, nseg, epi, st()))
    ms = timeit(fn)
    tf = 2.0 * M * N * K / (ms * 1e-3) / 1e12
    print(json.dumps(dict(op="gemm", M=M, N=N, K=K, epi=epi, nseg=nseg, ms=round(ms, 4), tflops=round(tf, 1))), flush=True)


def attn_case(B, H, N, hp):
    D = H * 64
    npad = (N + 63) // 64 * 64
    qk, qkl = rnd(B * N, 2 * D), rnd(B * N, 2 * D)
    vt, vtl = rnd(B * H, 64, npad), rnd(B * H, 64, npad)
    o, ol = torch.empty(B * N, D, dtype=torch.bfloat16, device=dev), torch.empty(B * N, D, dtype=torch.bfloat16, device=dev)
    fn = lambda: E.check(lib.f5_op_attention(P(qk), P(qkl), P(vt), P(vtl), P(o), P(ol), P(None), B, H, N, npad, D, C.c_float(0.125),
                                             hp, st()))
    ms = timeit(fn)
    tf = 4.0 * B * H * N * N * 64 / (ms * 1e-3) / 1e12
    print(json.dumps(dict(op="attention", B=B, H=H, N=N, hp=hp, ms=round(ms, 4), tflops=round(tf, 1))), flush=True)


def convpos_case(B, N, C_, nseg):
    x, xl = rnd(B * N, C_), rnd(B * N, C_)
    w, wl = rnd(C_, 31 * 64), rnd(C_, 31 * 64)
    bias = torch.zeros(C_, device=dev)
    o, ol = torch.empty(B * N, C_, dtype=torch.bfloat16, device=dev), torch.empty(B * N, C_, dtype=torch.bfloat16, device=dev)
    fn = lambda: E.check(lib.f5_op_convpos(P(x), P(xl), P(w), P(wl), P(bias), P(o), P(ol), P(None), B, N, C_, C_ // 64, 31, nseg, 0,
                                           st()))
    ms = timeit(fn)
    tf = 2.0 * B * N * 31 * 64 * C_ / (ms * 1e-3) / 1e12
    print(json.dumps(dict(op="convpos", B=B, N=N, C=C_, nseg=nseg, ms=round(ms, 4), tflops=round(tf, 1))), flush=True)


def ln_case(rows, dim):
    x = torch.randn(rows, dim, device=dev)
    sc, sh = torch.randn(dim, device=dev), torch.randn(dim, device=dev)
    hi, lo = torch.empty(rows, dim, dtype=torch.bfloat16, device=dev), torch.empty(rows, dim, dtype=torch.bfloat16, device=dev)
    fn = lambda: E.check(lib.f5_op_ln_modulate(P(x), P(sc), P(sh), P(hi), P(None), rows, dim, st()))
    ms = timeit(fn)
    gbs = rows * dim * 6 / (ms * 1e-3) / 1e9
    print(json.dumps(dict(op="ln_modulate", rows=rows, dim=dim, ms=round(ms, 4), GBps=round(gbs, 1))), flush=True)


if __name__ == "__main__":
    for tile in (2, 3, 5, 6):
        lib.f5_debug_set_gemm_tile(tile)
        for (N, K, epi) in ((3072, 1024, 1), (1024, 1024, 0), (2048, 1024, 2), (1024, 2048, 0)):
            print("tile", tile, end=" ")
            gemm_case(1874, N, K, epi, 1)
    for M in (7496, 59968):
        lib.f5_debug_set_gemm_tile(4)
        for (N, K, epi) in ((3072, 1024, 1), (1024, 1024, 0), (2048, 1024, 2), (1024, 2048, 0)):
            print("v2", end=" ")
            gemm_case(M, N, K, epi, 1)
    print("v2", end=" "); gemm_case(59968, 3072, 1024, 1, 3)
    print("v2", end=" "); gemm_case(8192, 8192, 8192, 1, 1)
    print("v2", end=" "); gemm_case(4096, 4096, 4096, 1, 1)
    lib.f5_debug_set_gemm_tile(0)
    for M in (1874, 7496, 59968):
        for (N, K, epi) in ((3072, 1024, 1), (1024, 1024, 0), (2048, 1024, 2), (1024, 2048, 0)):
            gemm_case(M, N, K, epi, 1)
    gemm_case(59968, 3072, 1024, 1, 3)
    gemm_case(8192, 8192, 8192, 1, 1)
    for ver in (1, 2):
        lib.f5_debug_set_attn_version(ver)
        print("attn version", ver)
        for B in (2, 64):
            attn_case(B, 16, 937, 0)
        attn_case(64, 16, 937, 1)
    for B in (2, 64):
        attn_case(B, 16, 937, 0)
    attn_case(2, 16, 937, 1)
    attn_case(8, 16, 4096, 0)
    for B in (2, 64):
        convpos_case(B, 937, 1024, 1)
    for rows in (1874, 59968):
        ln_case(rows, 1024)
