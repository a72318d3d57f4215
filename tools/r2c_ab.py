"""Round 2 same-call A/B of the residual-update GEMMs (out-proj / FF2 shapes, f16 operands unless told otherwise):
  * batch 1 (M = 1874, ring kernels): x / bias / gate / keep requested before the K loop (default) vs in the epilogue (flag 256);
  * batch 32 (M = 59 968, 256x256 kernel): x tile touched before the main loop -- 1/4 (512), 1/2 (1024), all lines (2048) -- vs
    not at all (0).
usage: python tools/r2c_ab.py [f16|bf16 ...]"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402
from tools.attn_prio_bench import graph_time  # noqa: E402

lib, dev, P = E.load_library(), torch.device("cuda:0"), E.ptr
D, FF, N = 1024, 2048, 937


def main(prec):
    dt = E.operand_dtype(prec)
    g = torch.Generator(device="cpu").manual_seed(0)
    for nb, arms in ((2, (("early", 0), ("late", 256))), (64, (("none", 0), ("quarter", 512), ("half", 1024), ("all", 2048)))):
        M = nb * N
        for (K, name) in ((D, "out-proj"), (FF, "ff2")):
            a = torch.randn(M, K, generator=g).to(dev).to(dt)
            w = (torch.randn(D, K, generator=g) * K ** -0.5).to(dev).to(dt)
            bias, gate, x = torch.zeros(D, device=dev), torch.full((D,), 0.01, device=dev), torch.zeros(M, D, device=dev)
            # a second, unrelated 245 MB stream between launches at batch 32 so that x is not trivially cache resident
            fn = lambda st: E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bias), P(gate), P(None), P(x), M, D, K, K, K,
                                                              D, 1, st))
            res = {nm: [] for nm, _ in arms}
            with E.operand_type(prec):
                for rnd in range(4):
                    for nm, fl in arms:
                        E.check(lib.f5_debug_set_gemm_flags(fl))
                        res[nm].append(graph_time(fn, reps=12 if nb > 2 else 44))
            E.check(lib.f5_debug_set_gemm_flags(0))
            print(json.dumps(dict(op=f"resid {name}", M=M, K=K, prec=prec, us={k: [round(v, 1) for v in vs] for k, vs in res.items()})),
                  flush=True)


if __name__ == "__main__":
    for prec in (sys.argv[1:] or ["f16"]):
        main(prec)
