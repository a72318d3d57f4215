"""Round 2 (second half) same-call A/B of the two changes to the block kernels, both operand builds, interleaved rounds:
  * gated residual update of the out-proj / FF2 GEMMs: L2 atomic add (gemm flag 8, experiment) vs load / add / store (default),
    at the batch-32 shapes (256x256 kernel) and the batch-1 shapes (ring kernels);
  * attention without the per-tile maximum (default) vs the previous kernels (variant bit 16), q plain or pre-multiplied by
    scale * log2(e), at batch 32 (large-grid kernel) and batch 1 (split-KV kernel).
usage: python tools/r2b_ab.py [f16|bf16 ...]"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402
from tools.attn_prio_bench import graph_time  # noqa: E402

lib, dev, P = E.load_library(), torch.device("cuda:0"), E.ptr
D, FF, H, N = 1024, 2048, 16, 937
npad = (N + 63) // 64 * 64
QPRE = 0.125 * 1.4426950408889634


def resid_ab(prec):
    dt = E.operand_dtype(prec)
    g = torch.Generator(device="cpu").manual_seed(0)
    for nb in (64, 2):
        M = nb * N
        for (K, name) in ((D, "out-proj"), (FF, "ff2")):
            a = torch.randn(M, K, generator=g).to(dev).to(dt)
            w = (torch.randn(D, K, generator=g) * K ** -0.5).to(dev).to(dt)
            bias, gate, x = torch.zeros(D, device=dev), torch.full((D,), 0.01, device=dev), torch.zeros(M, D, device=dev)
            fn = lambda st: E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bias), P(gate), P(None), P(x), M, D, K, K, K,
                                                              D, 1, st))
            res = {"atomic": [], "rmw": []}
            with E.operand_type(prec):
                for rnd in range(4):
                    for nm, fl in (("atomic", 8), ("rmw", 0)):
                        E.check(lib.f5_debug_set_gemm_flags(fl))
                        res[nm].append(graph_time(fn, reps=12 if nb > 2 else 44))
            E.check(lib.f5_debug_set_gemm_flags(0))
            fl = 2.0 * M * D * K
            print(json.dumps(dict(op=f"resid {name}", M=M, K=K, prec=prec, us={k: [round(v, 1) for v in vs] for k, vs in res.items()},
                                  tflops_best={k: round(fl / min(vs) / 1e6) for k, vs in res.items()})), flush=True)


def attn_ab(prec):
    dt = E.operand_dtype(prec)
    g = torch.Generator(device="cpu").manual_seed(0)
    for nb in (64, 2):
        q = torch.randn(nb * N, D, generator=g)
        k = torch.randn(nb * N, D, generator=g)
        qk = {False: torch.cat([q, k], 1).to(dev).to(dt), True: torch.cat([q * QPRE, k], 1).to(dev).to(dt)}
        vt = torch.zeros(nb * H, 64, npad, dtype=dt, device=dev)
        vt[..., :N] = torch.randn(nb * H, 64, N, generator=g).to(dev).to(dt)
        ao = torch.empty(nb * N, D, dtype=dt, device=dev)
        arms = [("old", 16, False), ("nomax", 0, False), ("nomax+premul", 0, True)]
        res = {a[0]: [] for a in arms}
        outs = {}
        with E.operand_type(prec):
            for rnd in range(4):
                for nm, bits, pre in arms:
                    E.check(lib.f5_debug_set_attn_variant(bits))
                    E.check(lib.f5_debug_set_op_q_premul(C.c_float(QPRE if pre else 0.0)))
                    fn = lambda st: E.check(lib.f5_op_attention(P(qk[pre]), P(None), P(vt), P(None), P(ao), P(None), P(None), nb, H, N,
                                                                npad, D, C.c_float(0.125), 0, st))
                    res[nm].append(graph_time(fn, reps=12 if nb > 2 else 44))
                    outs[nm] = ao.float().clone()
        E.check(lib.f5_debug_set_attn_variant(0))
        E.check(lib.f5_debug_set_op_q_premul(C.c_float(0.0)))
        fl = 4.0 * nb * H * N * N * 64
        print(json.dumps(dict(op="attention", nb=nb, prec=prec, us={k: [round(v, 1) for v in vs] for k, vs in res.items()},
                              tflops_best={k: round(fl / min(vs) / 1e6) for k, vs in res.items()},
                              max_diff_vs_old={k: float((v - outs["old"]).abs().max()) for k, v in outs.items()})), flush=True)


if __name__ == "__main__":
    for prec in (sys.argv[1:] or ["f16"]):
        resid_ab(prec)
        attn_ab(prec)
