"""A/B of the persistent walk of the 256x256 GEMM (f5_debug_set_gemm256_persist: one workgroup per CU, the next tile's first A halves
requested before the epilogue) on the four block GEMMs at the batch-32 shape: in-graph microseconds per launch, interleaved, and a
bitwise comparison of the outputs.  One JSON line per GEMM (profiles/r04/gemm256_persist_ab.jsonl).

    python tools/r4_persist_ab.py [--batch 32] [--flags 0]
"""
import argparse
import ctypes as C
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from f5_tts_mlx_amd import engine as E  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--flags", type=int, default=0, help="extra gemm debug flags for the persistent runs (32768 = no early A request)")
    ap.add_argument("--iters", type=int, default=20)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = E.load_library()
    P, st = E.ptr, (lambda: E.stream_ptr(dev))
    B, N, D, FF, H = a.batch, bench.N_FRAMES, 1024, 2048, 16
    M = 2 * B * N
    npad = (N + 63) // 64 * 64
    g = torch.Generator(device="cpu").manual_seed(0)
    opd = torch.float16
    mk = lambda std, *s: (torch.randn(*s, generator=g) * std).to(dev).to(opd)
    x1, x2 = mk(1.0, M, D), mk(1.0, M, FF)
    wq, wo, w1, w2 = mk(D ** -0.5, 3 * D, D), mk(D ** -0.5, D, D), mk(D ** -0.5, FF, D), mk(FF ** -0.5, D, FF)
    bq, b1, bd = (torch.randn(3 * D, generator=g) * 0.1).to(dev), (torch.randn(FF, generator=g) * 0.1).to(dev), (torch.randn(D, generator=g) * 0.1).to(dev)
    gate = torch.randn(D, generator=g).to(dev)
    keep = (torch.rand(M, generator=g) > 0.2).to(torch.uint8).to(dev)
    cos_t, sin_t = torch.empty(N, 32, device=dev), torch.empty(N, 32, device=dev)
    tt = [torch.empty(32, N, device=dev) for _ in range(4)]
    qpre = 0.125 * 1.4426950408889634
    with E.operand_type("f16"):
        E.check(lib.f5_op_rope_table(P(cos_t), P(sin_t), N, 64, st()))
        E.check(lib.f5_op_rope_table_t(P(tt[0]), P(tt[1]), P(tt[2]), P(tt[3]), N, 64, C.c_float(qpre), st()))
        E.check(lib.f5_debug_set_op_rope_tables_t(P(tt[0]), P(tt[1]), P(tt[2]), P(tt[3])))
        lib.f5_debug_set_op_q_premul(C.c_float(qpre))
        qk, vt = torch.zeros(M, 2 * D, dtype=opd, device=dev), torch.zeros(2 * B * H, 64, npad, dtype=opd, device=dev)
        ffh = torch.zeros(M, FF, dtype=opd, device=dev)
        xres = torch.zeros(M, D, device=dev)
        cases = {
            "out_proj": (1, lambda: E.check(lib.f5_op_gemm_resid_gate(P(x1), P(None), P(wo), P(None), P(bd), P(gate), P(keep), P(xres), M, D, D, D, D, D, 1, st())),
                         lambda: xres, 2.0 * M * D * D),
            "ff2": (1, lambda: E.check(lib.f5_op_gemm_resid_gate(P(x2), P(None), P(w2), P(None), P(bd), P(gate), P(keep), P(xres), M, D, FF, FF, FF, D, 1, st())),
                    lambda: xres, 2.0 * M * D * FF),
            "ff1": (2, lambda: E.check(lib.f5_op_gemm(P(x1), P(None), P(w1), P(None), P(b1), P(None), P(ffh), P(None), M, FF, D, D, D, FF, 1, 2, st())),
                    lambda: ffh, 2.0 * M * D * FF),
            "qkv": (4, lambda: E.check(lib.f5_op_qkv_rope(P(x1), P(None), P(wq), P(None), P(bq), P(cos_t), P(sin_t), P(qk), P(None), P(vt), P(None),
                                                          2 * B, N, npad, H, D, 1, st())),
                    lambda: torch.cat([qk.flatten(), vt.flatten()]), 2.0 * M * D * 3 * D),
        }
        try:
            for name, (bit, fn, out, flops) in cases.items():
                res, outs = {0: [], bit: []}, {}
                for rep in range(3):
                    for mask in (0, bit):
                        E.check(lib.f5_debug_set_gemm256_persist(mask))
                        E.check(lib.f5_debug_set_gemm_flags(a.flags if mask else 0))
                        xres.zero_()
                        fn()
                        torch.cuda.synchronize()
                        outs[mask] = out().clone()
                        res[mask].append(bench._time_launches(fn, dev, a.iters) * 1e3)
                off, on = min(res[0]), min(res[bit])
                print(json.dumps(dict(gemm=name, M=M, flags=a.flags, us_plain=round(off, 1), us_persistent=round(on, 1), speedup=round(off / on, 4),
                                      tflops_plain=round(flops / off / 1e6, 1), tflops_persistent=round(flops / on / 1e6, 1),
                                      all_plain=[round(v, 1) for v in res[0]], all_persistent=[round(v, 1) for v in res[bit]],
                                      bit_identical=bool(torch.equal(outs[0], outs[bit])))), flush=True)
        finally:
            E.check(lib.f5_debug_set_gemm256_persist(0))
            E.check(lib.f5_debug_set_gemm_flags(0))
            E.check(lib.f5_debug_set_op_rope_tables_t(P(None), P(None), P(None), P(None)))
            lib.f5_debug_set_op_q_premul(C.c_float(0.0))


if __name__ == "__main__":
    main()
