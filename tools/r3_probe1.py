"""Round-3 probe 1 (GPU box): where does a 256x256 GEMM launch at the B=32 shapes spend its time, and do two independent
half-batch sample() chains overlap when they run on two streams?

  (a) main loop only (gemm flag 1) and full launches at M = 65536 (whole rounds) for K = 1024 / 2048 / 4096:
      T(K) = rounds * (P + (K / 64) * s)  ->  per-tile fixed cost P (prologue + drain) and the per-K-tile slope s
  (b) the same at the real M = 59968 (tile quantisation: 11.02 / 7.34 / 3.67 rounds cost 12 / 8 / 4)
  (c) two GEMM chains of M = 29984 on two streams against one chain of M = 59968 (do kernels of two queues interleave?)
  (d) two B=16 engines sampling concurrently on two streams (optionally offset) against one B=32 engine
"""
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402

lib = E.load_library()
dev = torch.device("cuda:0")
P = E.ptr
OPD = torch.float16


def rnd(std, *s):
    return (torch.randn(*s, device=dev) * std).to(OPD)


def ev_time(fn, iters=10, warm=2, stream=None):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3   # us


def part_ab():
    out = []
    with E.operand_type("f16"):
        for M in (65536, 59968):
            for (N, K, kind) in ((1024, 1024, "resid"), (1024, 2048, "resid"), (1024, 4096, "resid"), (2048, 1024, "gelu"),
                                 (3072, 1024, "plain"), (3072, 4096, "plain")):
                a, w = rnd(1.0, M, K), rnd(K ** -0.5, N, K)
                bias, gate = torch.zeros(N, device=dev), torch.full((N,), 0.5, device=dev)
                x = torch.zeros(M, N, device=dev) if kind == "resid" else None
                oh = torch.empty(M, N, dtype=OPD, device=dev) if kind != "resid" else None
                st = lambda: E.stream_ptr(dev)
                if kind == "resid":
                    fn = lambda: E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bias), P(gate), P(None), P(x), M, N, K, K, K, N, 1, st()))
                else:
                    epi = 2 if kind == "gelu" else 1
                    fn = lambda: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(oh), P(None), M, N, K, K, K, N, 1, epi, st()))
                res = {}
                for flags in (0, 1):
                    lib.f5_debug_set_gemm_flags(flags)
                    res["full" if flags == 0 else "mainloop"] = round(ev_time(fn), 1)
                lib.f5_debug_set_gemm_flags(0)
                tiles = ((M + 255) // 256) * (N // 256)
                rec = dict(part="ab", M=M, N=N, K=K, kind=kind, tiles=tiles, rounds=round(tiles / 256, 2), us=res,
                           tf_full=round(2.0 * M * N * K / res["full"] / 1e6, 0), tf_ml=round(2.0 * M * N * K / res["mainloop"] / 1e6, 0))
                print(json.dumps(rec), flush=True)
                out.append(rec)
                del a, w, x, oh
    return out


def part_c():
    """two chains of 8 launches (out-proj shape, then FF1 shape alternating) on two streams vs one double-size chain"""
    with E.operand_type("f16"):
        D, FF = 1024, 2048

        def mk(M):
            return dict(M=M, a1=rnd(1.0, M, D), wo=rnd(D ** -0.5, D, D), w1=rnd(D ** -0.5, FF, D), w2=rnd(FF ** -0.5, D, FF),
                        bias=torch.zeros(FF, device=dev), gate=torch.full((D,), 0.5, device=dev), x=torch.zeros(M, D, device=dev),
                        ffh=torch.empty(M, FF, dtype=OPD, device=dev))

        def chain(b, s, reps=4, rot=0):
            sp = C.c_void_p(s.cuda_stream)
            M = b["M"]
            ops = [
                lambda: E.check(lib.f5_op_gemm_resid_gate(P(b["a1"]), P(None), P(b["wo"]), P(None), P(b["bias"]), P(b["gate"]), P(None), P(b["x"]), M, D, D, D, D, D, 1, sp)),
                lambda: E.check(lib.f5_op_gemm(P(b["a1"]), P(None), P(b["w1"]), P(None), P(b["bias"]), P(None), P(b["ffh"]), P(None), M, FF, D, D, D, FF, 1, 2, sp)),
                lambda: E.check(lib.f5_op_gemm_resid_gate(P(b["ffh"]), P(None), P(b["w2"]), P(None), P(b["bias"]), P(b["gate"]), P(None), P(b["x"]), M, D, FF, FF, FF, D, 1, sp)),
            ]
            for r in range(reps):
                for i in range(3):
                    ops[(i + rot) % 3]()

        big, h0, h1 = mk(59968), mk(29984), mk(29984)
        s0, s1 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)

        def timed(fn, iters=5):
            fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / iters * 1e3

        one = timed(lambda: chain(big, s0))
        seq = timed(lambda: (chain(h0, s0), chain(h1, s0)))
        par = timed(lambda: (chain(h0, s0), chain(h1, s1)))
        par_rot = timed(lambda: (chain(h0, s0, rot=0), chain(h1, s1, rot=1)))
        print(json.dumps(dict(part="c", what="12 GEMM launches (out-proj, FF1, FF2 shapes x4), ms", one_stream_M59968=round(one, 3),
                              halves_sequential=round(seq, 3), halves_two_streams=round(par, 3), halves_two_streams_rotated=round(par_rot, 3))), flush=True)


def part_d():
    from bench import synth_batch, N_FRAMES
    from f5_tts_mlx_amd.cfm import F5TTS
    from f5_tts_mlx_amd.dit import DiT
    from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights
    weights = synthetic_weights(F5TTS_335M, seed=42)
    models = []
    for _ in range(2):
        m = DiT.from_config(F5TTS_335M, precision="f16", device=dev)
        m.load_weights(weights)
        models.append(F5TTS(transformer=m))
    kw = dict(duration=N_FRAMES, steps=32, method="euler", cfg_strength=2.0, sway_sampling_coef=-1.0, use_graph=True)
    c32, t32, y32, _ = synth_batch(32, 0, dev)
    halves = [(c32[:16].contiguous(), t32[:16].contiguous(), y32[:16].contiguous()), (c32[16:].contiguous(), t32[16:].contiguous(), y32[16:].contiguous())]
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]

    def run32():
        return models[0].sample(c32, t32, y0=y32, **kw)[0]

    def run_half(i, s):
        with torch.cuda.stream(s):
            c, t, y = halves[i]
            return models[i].sample(c, t, y0=y, **kw)[0]

    def timed(fn, iters=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            r = fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3, r

    ms32, o32 = timed(run32)
    ms16, _ = timed(lambda: run_half(0, streams[0]))
    res = dict(part="d", b32_ms=round(ms32, 1), b16_alone_ms=round(ms16, 1))
    for delay_ms in (0, 0.3, 1.0):
        def both():
            a = run_half(0, streams[0])
            if delay_ms:
                with torch.cuda.stream(streams[1]):
                    torch.cuda._sleep(int(delay_ms * 1e-3 * 100e6))    # wall-clock ticks (100 MHz) on ROCm builds
            b = run_half(1, streams[1])
            return a, b
        ms, (oa, ob) = timed(both)
        res[f"two_b16_concurrent_delay{delay_ms}_ms"] = round(ms, 1)
    res["halves_equal_b32"] = bool(torch.equal(torch.cat([oa, ob]), o32))
    res["halves_vs_b32_maxabs"] = float((torch.cat([oa, ob]) - o32).abs().max())
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    which = sys.argv[1:] or ["ab", "c", "d"]
    if "ab" in which:
        part_ab()
    if "c" in which:
        part_c()
    if "d" in which:
        part_d()
