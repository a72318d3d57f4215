"""256x256 GEMM, 16-bit outputs at the batch-32 shapes: tile accumulated transposed + 8-byte LDS staging writes (default) vs the
straight order with 2-byte staging writes (gemm flag 16384); flag 1 = main loop only.  Same call, interleaved rounds.
usage: python tools/tr_epilogue_ab.py [f16|bf16]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402
from tools.attn_prio_bench import graph_time  # noqa: E402

lib, dev, P = E.load_library(), torch.device("cuda:0"), E.ptr
M = 59968


def main(prec):
    dt = E.operand_dtype(prec)
    g = torch.Generator(device="cpu").manual_seed(0)
    for (N, K, epi, name) in ((2048, 1024, 2, "ff1 gelu-tanh"), (3072, 1024, 1, "qkv-shaped plain 16-bit")):
        a = torch.randn(M, K, generator=g).to(dev).to(dt)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev).to(dt)
        bias = torch.zeros(N, device=dev)
        ob = torch.empty(M, N, dtype=dt, device=dev)
        fn = lambda st: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(ob), P(None), M, N, K, K, K, N, 1, epi, st))
        res = {}
        with E.operand_type(prec):
            for rnd in range(4):
                for nm, fl in (("transposed", 0), ("straight", 16384), ("mainloop_only", 1)):
                    E.check(lib.f5_debug_set_gemm_flags(fl))
                    res.setdefault(nm, []).append(round(graph_time(fn, reps=8), 1))
        E.check(lib.f5_debug_set_gemm_flags(0))
        fl = 2.0 * M * N * K
        print(json.dumps(dict(op=name, M=M, N=N, K=K, prec=prec, us=res, tflops_best={k: round(fl / min(v) / 1e6) for k, v in res.items()})),
              flush=True)


def qkv(prec):
    """QKV projection + RoPE + head split at batch 32: transposed q / k tiles (pair-major tables) vs straight tiles."""
    import ctypes as C
    dt = E.operand_dtype(prec)
    g = torch.Generator(device="cpu").manual_seed(1)
    D, H, N, nb = 1024, 16, 937, 64
    npad, Mq = 960, 64 * 937
    QPRE = 0.125 * 1.4426950408889634
    st0 = E.stream_ptr(dev)
    cos_t, sin_t = torch.empty((N, 32), device=dev), torch.empty((N, 32), device=dev)
    tt = [torch.empty((32, N), device=dev) for _ in range(4)]
    E.check(lib.f5_op_rope_table(P(cos_t), P(sin_t), N, 64, st0))
    E.check(lib.f5_op_rope_table_t(P(tt[0]), P(tt[1]), P(tt[2]), P(tt[3]), N, 64, C.c_float(QPRE), st0))
    E.check(lib.f5_debug_set_op_q_premul(C.c_float(QPRE)))
    x = torch.randn(Mq, D, generator=g).to(dev).to(dt)
    w = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).to(dt)
    bias = torch.zeros(3 * D, device=dev)
    qk = torch.empty(Mq, 2 * D, dtype=dt, device=dev)
    vt = torch.zeros(nb * H, 64, npad, dtype=dt, device=dev)
    fn = lambda st: E.check(lib.f5_op_qkv_rope(P(x), P(None), P(w), P(None), P(bias), P(cos_t), P(sin_t), P(qk), P(None), P(vt), P(None),
                                               nb, N, npad, H, D, 1, st))
    res = {}
    with E.operand_type(prec):
        for rnd in range(4):
            for nm, on in (("transposed_qk", True), ("straight", False)):
                E.check(lib.f5_debug_set_op_rope_tables_t(*([P(t) for t in tt] if on else [P(None)] * 4)))
                res.setdefault(nm, []).append(round(graph_time(fn, reps=8), 1))
    E.check(lib.f5_debug_set_op_rope_tables_t(P(None), P(None), P(None), P(None)))
    E.check(lib.f5_debug_set_op_q_premul(C.c_float(0.0)))
    fl = 2.0 * Mq * 3 * D * D
    print(json.dumps(dict(op="qkv + rope + head split", M=Mq, prec=prec, us=res, tflops_best={k: round(fl / min(v) / 1e6) for k, v in res.items()})),
          flush=True)


if __name__ == "__main__":
    qkv(sys.argv[1] if len(sys.argv) > 1 else "f16")
    main(sys.argv[1] if len(sys.argv) > 1 else "f16")
