"""A/B of the ring kernels' tile numbering (f5_debug_set_gemm_order: 2 = m fastest, the round-3 choice at batch 1; 0 = auto = band-major
for one-round launches whose A operand fits an L2) on sample() at batch 1, f16, 32-point Euler, graph replay, interleaved; plus the
op-level launch times of the three ring GEMMs of a block.  One JSON line (profiles/r04/ring_order_ab.json).

    python tools/r4_ring_order_ab.py [--reps 5]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from f5_tts_mlx_amd import engine as E  # noqa: E402
from f5_tts_mlx_amd.cfm import F5TTS  # noqa: E402
from f5_tts_mlx_amd.dit import DiT  # noqa: E402
from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    lib = E.load_library()
    m = DiT.from_config(F5TTS_335M, precision="f16", device=dev)
    m.load_weights(synthetic_weights(F5TTS_335M, seed=42))
    f5 = F5TTS(transformer=m)
    kw = dict(duration=bench.N_FRAMES, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, use_graph=True)
    cond, text, y0, _ = bench.synth_batch(1, 0, dev)
    res, outs, ops = {2: [], 0: []}, {}, {}
    try:
        for rep in range(a.reps + 1):
            for order in (2, 0):
                E.check(lib.f5_debug_set_gemm_order(order))
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                out, _ = f5.sample(cond, text, y0=y0, **kw)
                torch.cuda.synchronize()
                if rep:
                    res[order].append((time.perf_counter() - t0) * 1e3)
                outs[order] = out
        for order in (2, 0, 3, 1, 2, 0, 3, 1):
            E.check(lib.f5_debug_set_gemm_order(order))
            ks, _ = bench.kernel_rooflines("f16", dev, 1, iters=50, peak_meas=dict(workload_like_operands=1.0, constant_operands=1.0))
            cur = {k["key"]: round(k["avg_launch_ms"] * 1e3, 2) for k in ks}
            ops[order] = {k: min(v, ops[order][k]) for k, v in cur.items()} if order in ops else cur
    finally:
        E.check(lib.f5_debug_set_gemm_order(0))
    print(json.dumps(dict(B=1, precision="f16", ms_m_fastest=round(min(res[2]), 3), ms_auto_band_major=round(min(res[0]), 3),
                          all_m_fastest=[round(v, 2) for v in res[2]], all_auto=[round(v, 2) for v in res[0]],
                          bit_identical=bool(torch.equal(outs[0], outs[2])), op_us_m_fastest=ops[2], op_us_auto=ops[0],
                          op_us_band_major_forced=ops[3], op_us_n_fastest=ops[1])))


if __name__ == "__main__":
    main()
