"""B=32 block GEMM shapes: the 256x256 kernel (one workgroup per CU) against the 128x256 two-per-CU kernel under the three
issue-priority schemes (0 MFMA clusters, 1 none, 2 epilogue), two stagger values, interleaved rounds in one process."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402
from tools.attn_prio_bench import graph_time  # noqa: E402

lib, dev, P = E.load_library(), torch.device("cuda:0"), E.ptr
M = 59968


def main():
    prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
    dt = E.operand_dtype(prec)
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda std, *s: (torch.randn(*s, generator=g) * std).to(dev).to(dt)
    for (N, K, epi, name) in ((3072, 1024, 1, "qkv-shape 16-bit out"), (2048, 1024, 2, "ff1 gelu"), (1024, 1024, 4, "oproj resid"),
                              (1024, 2048, 4, "ff2 resid")):
        a, w = mk(1.0, M, K), mk(K ** -0.5, N, K)
        bias = torch.zeros(N, device=dev); gate = torch.ones(N, device=dev) * 0.01
        of = torch.zeros(M, N, device=dev); ob = torch.empty(M, N, dtype=dt, device=dev)
        if epi == 4:
            fn = lambda st: E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bias), P(gate), P(None), P(of), M, N, K, K, K, N, 1, st))
        else:
            fn = lambda st: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(ob), P(None), M, N, K, K, K, N, 1, epi, st))
        cases = [("256x256", 2, 0, -1)] + [(f"128x256 prio{p} stagger{sg}", 3, p, sg) for p in (0, 1, 2) for sg in (-1, 0)]
        row = {c[0]: [] for c in cases}
        with E.operand_type(prec):
            for rnd in range(3):
                for nm, big, prio, sg in cases:
                    E.check(lib.f5_debug_set_gemm_big_kernel(big, sg))
                    E.check(lib.f5_debug_set_gemm_v3_prio(prio))
                    row[nm].append(round(graph_time(fn, reps=8, iters=4), 1))
        E.check(lib.f5_debug_set_gemm_big_kernel(2, -1)); E.check(lib.f5_debug_set_gemm_v3_prio(0))
        print(json.dumps(dict(op=name, prec=prec, us=row, tflops_best={k: round(2.0 * M * N * K / min(v) / 1e6) for k, v in row.items()})), flush=True)


if __name__ == "__main__":
    main()
