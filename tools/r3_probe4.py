"""Round-3 probe 4: is the 256x256 GEMM main loop power (DVFS) limited?  Same launch on random, constant and zero operands."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E
from tools.r3_probe2 import ev_time, setk, lib, dev, P, st
M, D = 59968, 1024
opd = torch.float16
with E.operand_type("f16"):
    for K in (1024, 2048):
        fills = {
            "random N(0,1) x N(0,1/K)": (torch.randn(M, K, device=dev).to(opd), (torch.randn(D, K, device=dev) * K ** -0.5).to(opd)),
            "zeros": (torch.zeros(M, K, dtype=opd, device=dev), torch.zeros(D, K, dtype=opd, device=dev)),
            "constant 0.5 / 0.03": (torch.full((M, K), 0.5, dtype=opd, device=dev), torch.full((D, K), 0.03, dtype=opd, device=dev)),
            "random sign, constant magnitude": ((torch.randint(0, 2, (M, K), device=dev) * 2 - 1).to(opd), ((torch.randint(0, 2, (D, K), device=dev) * 2 - 1) * 0.03).to(opd)),
        }
        bd, gate = torch.zeros(D, device=dev), torch.full((D,), 0.5, device=dev)
        x = torch.zeros(M, D, device=dev)
        for kern in (2, 5):
            setk(kern)
            for name, (a, w) in fills.items():
                fn = lambda: E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bd), P(gate), P(None), P(x), M, D, K, K, K, D, 1, st()))
                rec = dict(kernel=kern, K=K, fill=name)
                for flags in (1, 0):
                    lib.f5_debug_set_gemm_flags(flags)
                    us = ev_time(fn, iters=20, warm=5)
                    rec["ml" if flags else "full"] = [round(us, 1), round(2.0 * M * D * K / us / 1e6)]
                lib.f5_debug_set_gemm_flags(0)
                print(json.dumps(rec), flush=True)
        del fills
setk(2)
