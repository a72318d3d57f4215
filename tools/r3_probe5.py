"""Round-3 probe 5 (GPU box): the role-split 128x256 kernel (gemm_rs128.hip, tile 14) at the batch-1 shapes (M = 1874) against the
shipped small-M kernels: bitwise equality (+ repeat runs) and in-graph timing of QKV / FF1 / out-proj / FF2."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402

lib = E.load_library()
dev = torch.device("cuda:0")
P = E.ptr
st = lambda: E.stream_ptr(dev)
opd = torch.float16


def graph_time(fn, iters=40):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(iters):
            fn()
    g.replay()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        g.replay()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


def main():
    M, D, FF, H, n, B = 1874, 1024, 2048, 16, 937, 2
    npad = 960
    g = torch.Generator(device="cpu").manual_seed(3)
    mk = lambda std, *s: (torch.randn(*s, generator=g) * std).to(dev).to(opd)
    x1, x2 = mk(1.0, M, D), mk(1.0, M, FF)
    wq, wo, w1, w2 = mk(D ** -0.5, 3 * D, D), mk(D ** -0.5, D, D), mk(D ** -0.5, FF, D), mk(FF ** -0.5, D, FF)
    bq, b1, bd = (torch.randn(3 * D, generator=g) * 0.1).to(dev), (torch.randn(FF, generator=g) * 0.1).to(dev), (torch.randn(D, generator=g) * 0.1).to(dev)
    gate = (torch.randn(D, generator=g) * 0.5).to(dev)
    cos_t, sin_t = torch.rand(n, 32, generator=g).to(dev), torch.rand(n, 32, generator=g).to(dev)
    tt = [torch.empty(32, n, device=dev) for _ in range(4)]
    x0 = torch.randn(M, D, generator=g).to(dev)
    with E.operand_type("f16"):
        E.check(lib.f5_op_rope_table_t(P(tt[0]), P(tt[1]), P(tt[2]), P(tt[3]), n, 64, C.c_float(0.18), st()))
        E.check(lib.f5_debug_set_op_rope_tables_t(P(tt[0]), P(tt[1]), P(tt[2]), P(tt[3])))
        lib.f5_debug_set_op_q_premul(C.c_float(0.18))

        def qkv():
            qk = torch.zeros(M, 2 * D, dtype=opd, device=dev)
            vt = torch.zeros(B * H, 64, npad, dtype=opd, device=dev)
            E.check(lib.f5_op_qkv_rope(P(x1), P(None), P(wq), P(None), P(bq), P(cos_t), P(sin_t), P(qk), P(None), P(vt), P(None), B, n, npad, H, D, 1, st()))
            return qk, vt

        def ff1():
            o = torch.zeros(M, FF, dtype=opd, device=dev)
            E.check(lib.f5_op_gemm(P(x1), P(None), P(w1), P(None), P(b1), P(None), P(o), P(None), M, FF, D, D, D, FF, 1, 2, st()))
            return (o,)

        def out():
            x = x0.clone()
            E.check(lib.f5_op_gemm_resid_gate(P(x1), P(None), P(wo), P(None), P(bd), P(gate), P(None), P(x), M, D, D, D, D, D, 1, st()))
            return (x,)

        def ff2():
            x = x0.clone()
            E.check(lib.f5_op_gemm_resid_gate(P(x2), P(None), P(w2), P(None), P(bd), P(gate), P(None), P(x), M, D, FF, FF, FF, D, 1, st()))
            return (x,)

        # persistent outputs for timing (no allocation inside the graph)
        qk_t, vt_t = torch.zeros(M, 2 * D, dtype=opd, device=dev), torch.zeros(B * H, 64, npad, dtype=opd, device=dev)
        o_t, x_t = torch.zeros(M, FF, dtype=opd, device=dev), x0.clone()
        timed = {
            "qkv": lambda: E.check(lib.f5_op_qkv_rope(P(x1), P(None), P(wq), P(None), P(bq), P(cos_t), P(sin_t), P(qk_t), P(None), P(vt_t), P(None), B, n, npad, H, D, 1, st())),
            "ff1": lambda: E.check(lib.f5_op_gemm(P(x1), P(None), P(w1), P(None), P(b1), P(None), P(o_t), P(None), M, FF, D, D, D, FF, 1, 2, st())),
            "out": lambda: E.check(lib.f5_op_gemm_resid_gate(P(x1), P(None), P(wo), P(None), P(bd), P(gate), P(None), P(x_t), M, D, D, D, D, D, 1, st())),
            "ff2": lambda: E.check(lib.f5_op_gemm_resid_gate(P(x2), P(None), P(w2), P(None), P(bd), P(gate), P(None), P(x_t), M, D, FF, FF, FF, D, 1, st())),
        }
        fns = {"qkv": qkv, "ff1": ff1, "out": out, "ff2": ff2}
        flops = {"qkv": 2.0 * M * D * 3 * D, "ff1": 2.0 * M * D * FF, "out": 2.0 * M * D * D, "ff2": 2.0 * M * FF * D}
        for name in ("qkv", "ff1", "out", "ff2"):
            rec = dict(op=name)
            E.check(lib.f5_debug_set_gemm_tile(13 if name == "qkv" else 0))      # QKV: compare among the transposed-tile kernels
            ref = fns[name]()
            torch.cuda.synchronize()
            for tile in ((0, 13, 14) if name == "qkv" else (0, 14)):
                E.check(lib.f5_debug_set_gemm_tile(tile))
                got = fns[name]()
                torch.cuda.synchronize()
                eq = all(torch.equal(a, b) for a, b in zip(ref, got))
                rep = True
                for _ in range(8):
                    again = fns[name]()
                    torch.cuda.synchronize()
                    rep = rep and all(torch.equal(a, b) for a, b in zip(got, again))
                us = graph_time(timed[name])
                lib.f5_debug_set_gemm_flags(1)
                us_ml = graph_time(timed[name])
                lib.f5_debug_set_gemm_flags(0)
                rec[f"tile{tile}"] = dict(us=round(us, 2), tf=round(flops[name] / us / 1e6), ml_us=round(us_ml, 2), bitwise_equal_default=eq, repeatable=rep)
            E.check(lib.f5_debug_set_gemm_tile(0))
            print(json.dumps(rec), flush=True)
        lib.f5_debug_set_op_q_premul(C.c_float(0.0))
        E.check(lib.f5_debug_set_op_rope_tables_t(P(None), P(None), P(None), P(None)))


if __name__ == "__main__":
    main()
