"""Round 6: is the tile count's remainder over 256 CUs paid for?  (GPU box, measurement only.)

The block GEMMs run 256 x 256 tiles, one workgroup per CU.  At M = 59 968 (235 row panels, the last one 64 rows) that is 940 / 1880 /
2820 tiles = 3.67 / 7.34 / 11.02 rounds of 256 CUs.  If the launches cost whole rounds, 8 % of each is tail.  This sweeps M (batch
count for QKV, whose rows are batch x 937) across the round boundaries and prints us per launch: a staircase says "quantised", a line
through the origin says "the tail is already cheap".

usage: python tools/r6_m_sweep.py > gpurun_out/TAG/m_sweep.jsonl
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402
from tools.yardstick import ev_time  # noqa: E402

lib = E.load_library()
dev = torch.device("cuda:0")
P = E.ptr
st = lambda: E.stream_ptr(dev)   # noqa: E731
D, FF, H, NF = 1024, 2048, 16, 937
opd = torch.float16
g = torch.Generator(device="cpu").manual_seed(0)
mk = lambda std, *s: (torch.randn(*s, generator=g) * std).to(dev).to(opd)   # noqa: E731


def sweep(name, N, K, kind, batches):
    w = mk(K ** -0.5, N, K)
    bias = torch.zeros(N, device=dev)
    gate = torch.full((N,), 0.5, device=dev)
    bmax = max(batches)
    a_all = mk(1.0, bmax * NF, K)
    xres = torch.zeros(bmax * NF, D, device=dev)
    out16 = torch.empty(bmax * NF, N, dtype=opd, device=dev)
    cos_t, sin_t = torch.ones(NF, 32, device=dev), torch.zeros(NF, 32, device=dev)
    npad = (NF + 63) // 64 * 64
    qk = torch.empty(bmax * NF, 2 * D, dtype=opd, device=dev)
    vt = torch.zeros(bmax * H, 64, npad, dtype=opd, device=dev)
    with E.operand_type("f16"):
        for rnd in range(2):
            for b in batches:
                M = b * NF
                a = a_all[:M]
                if kind == "resid":
                    fn = lambda: E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bias), P(gate), P(None), P(xres),   # noqa: E731
                                                                   M, D, K, K, K, D, 1, st()))
                elif kind == "gelu":
                    fn = lambda: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(out16), P(None), M, N, K, K, K, N, 1, 2, st()))   # noqa: E731
                else:
                    fn = lambda: E.check(lib.f5_op_qkv_rope(P(a), P(None), P(w), P(None), P(bias), P(cos_t), P(sin_t), P(qk), P(None), P(vt),   # noqa: E731
                                                            P(None), b, NF, npad, H, D, 1, st()))
                us = ev_time(fn, iters=20)
                panels = (M + 255) // 256
                tiles = panels * (N // 256)
                print(json.dumps(dict(shape=name, round=rnd, batch=b, M=M, panels=panels, last_panel_rows=M - (panels - 1) * 256, tiles=tiles,
                                      rounds_of_256=round(tiles / 256, 3), us=round(us, 1), us_per_tile_round=round(us / (tiles / 256), 2),
                                      tflops=round(2.0 * M * N * K / us / 1e6))), flush=True)


if __name__ == "__main__":
    bs = [48, 52, 54, 55, 56, 58, 60, 62, 63, 64, 65, 66, 68, 69, 70, 72]
    which = sys.argv[1:] or ["qkv", "out_proj", "ff1", "ff2"]
    for name, N, K, kind in (("qkv", 3 * D, D, "qkv"), ("out_proj", D, D, "resid"), ("ff1", FF, D, "gelu"), ("ff2", D, FF, "resid")):
        if name in which:
            sweep(name, N, K, kind, bs)
