"""conv-pos kernel (dit.py:29-50) at the batch-1 bench shape (2 x 937 tokens, 1024 channels, 16 groups, 31 taps) and at batch 32:
taps per pipeline step x block numbering, in-graph microseconds per launch."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402
from tools.attn_prio_bench import graph_time  # noqa: E402

lib, dev, P = E.load_library(), torch.device("cuda:0"), E.ptr


def main():
    for nb in (2, 64):
        N, C, G, taps = 937, 1024, 16, 31
        dt = E.operand_dtype("f16")
        g = torch.Generator(device="cpu").manual_seed(0)
        x = torch.randn(nb * N, C, generator=g).to(dev).to(dt)
        w = (torch.randn(C, taps * 64, generator=g) * (taps * 64) ** -0.5).to(dev).to(dt)
        bias = torch.zeros(C, device=dev)
        out = torch.empty(nb * N, C, dtype=dt, device=dev)
        acc = torch.zeros(nb * N, C, device=dev)
        res = {}
        with E.operand_type("f16"):
            for mode, o, a in ((0, out, None), (1, None, acc)):
                fn = lambda st: E.check(lib.f5_op_convpos(P(x), P(None), P(w), P(None), P(bias), P(o), P(None), P(a), nb, N, C, G, taps, 1,
                                                          mode, st))
                for xcd in (0, 1):
                    for tps in (1, 2, 4):
                        E.check(lib.f5_debug_set_convpos_xcd_map(xcd))
                        E.check(lib.f5_debug_set_convpos_tps(tps))
                        res[f"mode{mode}.xcd{xcd}.tps{tps}"] = round(min(graph_time(fn) for _ in range(3)), 1)
        E.check(lib.f5_debug_set_convpos_xcd_map(1))
        E.check(lib.f5_debug_set_convpos_tps(0))
        fl = 2.0 * nb * N * taps * 64 * C
        print(json.dumps(dict(batch_rows=nb, us=res, gflop=round(fl / 1e9, 2))), flush=True)


if __name__ == "__main__":
    main()
