#!/usr/bin/env python
"""bench.py -- F5-TTS 335M flow-matching sampling throughput on MI355X (BASELINE.json metric).

A "step" = one `F5TTS.sample()` call (32-point Euler = 31 updates = 62 DiT forwards with CFG) over a
batch of synthetic 10 s utterances (N = 937 mel frames, SURVEY.md §8(d) inputs), inputs resident in
HBM, output = final mel on device.  value = mel frames produced per second, whole job.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--precision bf16|bf16x3]

Multi-GPU (utterances are independent): one process per GPU under torch.distributed.run,
weights generated on rank 0 and replicated with ONE RCCL broadcast of the weights arena, every rank
samples its own B utterances, no data-path collective ("scaling": "weak").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from f5_tts_mlx_amd.audio import log_mel_spectrogram  # noqa: E402
from f5_tts_mlx_amd.cfm import F5TTS, time_grid  # noqa: E402
from f5_tts_mlx_amd.dit import DiT  # noqa: E402
from f5_tts_mlx_amd.weights import F5TTS_335M, synthetic_weights  # noqa: E402

N_FRAMES = 937            # int(10.0 * 93.75), generate.py:23,163
REF_SAMPLES = 72_000      # 3.0 s reference audio -> 281 mel frames
NT = 160
ODE_POINTS = 32
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, MI355X_MICROARCH.md


def flops_forward(N: int, hoisted: bool) -> float:
    """Algorithmic flops of one DiT utterance-forward, SURVEY.md §8(d).  hoisted=True drops the work the engine
    runs once per sample instead of once per forward (text path, adaLN/time linears, cond/text input proj)."""
    blocks = 22 * (N * 2 * (4 * 1024 ** 2 + 2 * 1024 * 2048) + 4 * N ** 2 * 1024)
    conv = 2 * N * 2 * 31 * 64 * 1024
    out = N * 2 * 1024 * 100
    if hoisted:
        return blocks + conv + out + N * 2 * 128 * 1024
    ada = 22 * 2 * 1024 * 6144 + 2 * 1024 * 2048 + 2 * (256 * 1024 + 1024 ** 2)
    text = 4 * N * (2 * 2 * 512 * 1024 + 2 * 7 * 512)
    return blocks + conv + out + N * 2 * 712 * 1024 + ada + text


def synth_batch(B: int, first: int, device):
    conds, texts, y0s = [], [], []
    for i in range(first, first + B):
        wave = (np.random.default_rng(1234 + i).standard_normal(REF_SAMPLES).astype(np.float32) * np.float32(0.1))
        conds.append(log_mel_spectrogram(torch.from_numpy(wave).to(device))[0])
        texts.append(np.random.default_rng(2345 + i).integers(0, 2545, NT).astype(np.int32))
        y0s.append(np.random.default_rng(3456 + i).standard_normal((100, N_FRAMES)).astype(np.float32).T)
    cond = torch.stack(conds)                                            # (B, 281, 100) on device
    text = torch.from_numpy(np.stack(texts)).to(device)
    y0 = torch.from_numpy(np.ascontiguousarray(np.stack(y0s))).to(device)
    return cond, text, y0


def _time_launches(run, dev, iters: int) -> float:
    """Average duration (ms) of `iters` back-to-back launches of `run`, replayed from a hipGraph (as the kernel runs inside
    sample(): no host launch gaps), timed with HIP events on the stream the graph is launched on."""
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    side = torch.cuda.Stream(device=dev)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(iters):
            run()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def gemm_roofline(model: DiT, B: int, iters: int = 20):
    """Live HIP-event timing of the dominant kernel (QKV projection GEMM, f5_gemm_kernel<EPI_QKV_ROPE>) at the
    bench shape: M = 2*B*N rows (cond + null), K = 1024, N = 3072."""
    from f5_tts_mlx_amd import engine as E
    lib, dev = E.load_library(), model.device
    M, D, H = 2 * B * N_FRAMES, 1024, 16
    npad = (N_FRAMES + 63) // 64 * 64
    nseg = 3 if model.precision == "bf16x3" else 1
    g = torch.Generator(device="cpu").manual_seed(0)
    if model.precision == "mxfp8":
        return gemm_roofline_f8(lib, E, dev, M, D, g, iters)
    # operand statistics of the real workload (activations ~N(0,1), weights ~N(0,1/fan_in)): data toggling sets the
    # DVFS clock, so a microbenchmark on hotter random data would not agree with the in-graph rocprof average
    mk = lambda std, *s: (torch.randn(*s, generator=g) * std).to(dev).to(torch.bfloat16)
    a_hi, a_lo, w_hi, w_lo = mk(1.0, M, D), mk(0.004, M, D), mk(D ** -0.5, 3 * D, D), mk(1e-4, 3 * D, D)
    bias = torch.zeros(3 * D, device=dev)
    cos_t, sin_t = torch.ones(N_FRAMES, 32, device=dev), torch.zeros(N_FRAMES, 32, device=dev)
    qk = [torch.empty(M, 2 * D, dtype=torch.bfloat16, device=dev) for _ in range(2)]
    vt = [torch.zeros(2 * B * H, 64, npad, dtype=torch.bfloat16, device=dev) for _ in range(2)]
    lo = (lambda t: t) if nseg == 3 else (lambda t: None)      # plain bf16 mode has no "lo" operands / outputs
    def run():
        E.check(lib.f5_op_qkv_rope(E.ptr(a_hi), E.ptr(lo(a_lo)), E.ptr(w_hi), E.ptr(lo(w_lo)), E.ptr(bias), E.ptr(cos_t),
                                   E.ptr(sin_t), E.ptr(qk[0]), E.ptr(lo(qk[1])), E.ptr(vt[0]), E.ptr(lo(vt[1])), 2 * B, N_FRAMES,
                                   npad, H, D, nseg, E.stream_ptr(dev)))
    ms = _time_launches(run, dev, iters)
    flops = 2.0 * M * D * 3 * D                      # algorithmic: one (hi*hi) pass, whatever the precision mode
    achieved = flops / (ms * 1e-3) / 1e12
    shape = f"M={M} N={3 * D} K={D}"
    traffic = None
    try:   # HBM/fabric bytes per launch from the last committed rocprofv3 --pmc pass (profiles/pmc_traffic.json), if it is this shape
        pm = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))["qkv_gemm"]
        if pm.get("shape") == shape and pm.get("precision") == model.precision:
            traffic = pm["fetch_bytes_corrected_x2"] + pm["write_bytes"]
    except Exception:
        pass
    return dict(bound="mfma", kernel="QKV projection GEMM + bias + RoPE + head split (f5_gemm*_kernel<EPI_QKV_ROPE>)", shape=shape,
                avg_launch_ms=ms, achieved=achieved, peak=BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=achieved / BF16_PEAK_TFLOPS,
                traffic=traffic, traffic_unit="bytes/launch (FETCH_SIZE x2 gfx950 correction + WRITE_SIZE)",
                algorithmic_bytes=2 * M * D + 2 * 3 * D * D + 2 * M * 3 * D)


FP8_PEAK_TFLOPS = 5000.0   # dense fp8 (MX) MFMA peak, MI355X_MICROARCH.md


def gemm_roofline_f8(lib, E, dev, M, D, g, iters):
    """mxfp8 mode: the QKV projection runs on f5_gemm256f8_kernel (MX-fp8 operands, v_mfma_scale_f32_32x32x64_f8f6f4).  Timed
    here through f5_op_gemm_f8 with the plain bf16-output epilogue (same main loop and output bytes; the RoPE / head-split
    epilogue of the in-engine launch is not exported as an op), operands quantised from workload-like data."""
    a = (torch.randn(M, D, generator=g)).to(dev)
    w = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev)
    a8, asc = torch.empty(M, D, dtype=torch.uint8, device=dev), torch.empty(M, D // 32, dtype=torch.uint8, device=dev)
    w8, wsc = torch.empty(3 * D, D, dtype=torch.uint8, device=dev), torch.empty(3 * D, D // 32, dtype=torch.uint8, device=dev)
    st = E.stream_ptr(dev)
    E.check(lib.f5_op_quantize_mx(E.ptr(a), D, E.ptr(a8), D, E.ptr(asc), M, D, st))
    E.check(lib.f5_op_quantize_mx(E.ptr(w), D, E.ptr(w8), D, E.ptr(wsc), 3 * D, D, st))
    bias = torch.zeros(3 * D, device=dev)
    out = torch.empty(M, 3 * D, dtype=torch.bfloat16, device=dev)
    def run():
        E.check(lib.f5_op_gemm_f8(E.ptr(a8), E.ptr(asc), E.ptr(w8), E.ptr(wsc), E.ptr(bias), E.ptr(None), E.ptr(None), E.ptr(None),
                                  E.ptr(out), E.ptr(None), E.ptr(None), M, 3 * D, D, D, D, 3 * D, 1, E.stream_ptr(dev)))
    ms = _time_launches(run, dev, iters)
    achieved = 2.0 * M * D * 3 * D / (ms * 1e-3) / 1e12
    return dict(bound="mfma", kernel="QKV-shaped MX-fp8 GEMM + bias, bf16 out (f5_gemm256f8_kernel<EPI_BF16>)", shape=f"M={M} N={3 * D} K={D}",
                avg_launch_ms=ms, achieved=achieved, peak=FP8_PEAK_TFLOPS, unit="TFLOP/s", frac=achieved / FP8_PEAK_TFLOPS,
                traffic=None, traffic_unit="bytes/launch", algorithmic_bytes=M * D + 3 * D * D + (M + 3 * D) * D // 32 + 2 * M * 3 * D)


def _cpu_model() -> str:
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(weights, budget_s: float = 25.0):
    """The oracle (CPU restatement of the reference, kind="port") timed on the host cores on a bounded sample: full-size
    fp32 DiT forwards at N=937 (B=1), first a short sweep over thread counts (one forward each), then the best setting is
    timed again; extrapolated to the 62 forwards of a 32-point Euler solve.  ~10-30 s of CPU work in total."""
    from oracle import f5_oracle as O   # checker / baseline leg only
    orc = O.DiTOracle(F5TTS_335M, weights)
    r = np.random.default_rng(0)
    x = torch.from_numpy(r.standard_normal((1, N_FRAMES, 100)).astype(np.float32))
    cond = torch.zeros((1, N_FRAMES, 100))
    cond[:, :281] = torch.from_numpy(r.standard_normal((1, 281, 100)).astype(np.float32))
    text = torch.from_numpy(r.integers(0, 2545, (1, NT)).astype(np.int32))
    ncpu = os.cpu_count() or 1
    cands = [c for c in (8, 16, 32, 64) if c <= ncpu] or [ncpu]   # BLAS at these sizes stops scaling (and collapses) beyond ~32 threads
    t_start = time.perf_counter()
    best, sweep = None, {}
    orc.forward(x[:, :256], cond[:, :256], text, torch.tensor(0.3), False, False, None)    # warm the allocator / thread pool
    for c in cands:
        torch.set_num_threads(c)
        t0 = time.perf_counter()
        orc.forward(x, cond, text, torch.tensor(0.3), False, False, None)
        sweep[c] = time.perf_counter() - t0
        if best is None or sweep[c] < sweep[best]:
            best = c
        if sweep[c] > 1.25 * sweep[best] or time.perf_counter() - t_start > budget_s:
            break
    torch.set_num_threads(best)
    n_rep, t0 = 0, time.perf_counter()
    while n_rep < 2 or (time.perf_counter() - t_start < budget_s and n_rep < 4):
        orc.forward(x, cond, text, torch.tensor(0.3), bool(n_rep % 2), bool(n_rep % 2), None)
        n_rep += 1
    dt = (time.perf_counter() - t0) / n_rep
    n_fwd = 2 * (ODE_POINTS - 1)
    return dict(value=N_FRAMES / (n_fwd * dt), unit="mel-frames/s", cores=best, kind="port",
                sample=f"{n_rep} full-size fp32 DiT forwards (B=1, N={N_FRAMES}) of the oracle on torch-CPU with {best} threads "
                       f"(sweep s/forward: {dict((k, round(v, 2)) for k, v in sweep.items())}), {dt:.2f} s each, "
                       f"extrapolated x{n_fwd} forwards per utterance",
                rtf=10.0 / (n_fwd * dt), host_cpus=ncpu, cpu_model=_cpu_model())


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=1, help="utterances per GPU (BASELINE configs[1] = 1, configs[2] = 32)")
    ap.add_argument("--precision", default="bf16", choices=["bf16", "bf16x3", "mxfp8"])
    ap.add_argument("--vocoder", action="store_true", help="include the Vocos vocoder (random-init) in the timed region: mel -> waveform")
    ap.add_argument("--config", default=None, choices=["c5"],
                    help="c5 = BASELINE configs[4]: MX-fp8 block GEMMs + Vocos, 16-point midpoint, batch 32")
    ap.add_argument("--method", default="euler")
    ap.add_argument("--ode-points", type=int, default=ODE_POINTS)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()
    if args.config == "c5":
        args.precision, args.vocoder, args.method, args.ode_points = "mxfp8", True, "midpoint", 16
        if args.batch == 1:
            args.batch = 32

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist_on = world > 1
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device(f"cuda:{local}"))
    device = torch.device(f"cuda:{local}")
    torch.cuda.set_device(device)
    B = args.batch

    model = DiT.from_config(F5TTS_335M, precision=args.precision, device=device)
    weights = None
    t_w = time.perf_counter()
    if rank == 0:
        weights = synthetic_weights(F5TTS_335M, seed=42)
        model.load_weights(weights)
    if dist_on:
        from f5_tts_mlx_amd.dist import broadcast_weights
        bcast_ms = broadcast_weights(model.engine, src=0)
    else:
        bcast_ms = None
    load_s = time.perf_counter() - t_w

    vocoder = None
    if args.vocoder:
        from f5_tts_mlx_amd.vocos import Vocos, synthetic_vocos_weights
        vocoder = Vocos(synthetic_vocos_weights(seed=7), device=device).decode
    f5 = F5TTS(transformer=model, vocoder=vocoder)
    cond, text, y0 = synth_batch(B, first=rank * B, device=device)
    kw = dict(duration=N_FRAMES, steps=args.ode_points, method=args.method, cfg_strength=2.0, sway_sampling_coef=-1.0, y0=y0,
              use_graph=not args.no_graph)

    def barrier():
        torch.cuda.synchronize()
        if dist_on:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out, _ = f5.sample(cond, text, **kw)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out, _ = f5.sample(cond, text, **kw)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist_on:
        tt = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())
    assert torch.isfinite(out).all()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        frames = world * B * N_FRAMES
        value = frames / (ms_per_step * 1e-3)
        per = {"euler": 1, "midpoint": 2, "rk4": 4}[args.method]
        n_fwd = 2 * per * (args.ode_points - 1)
        exec_tflop = world * B * n_fwd * flops_forward(N_FRAMES, hoisted=True) / 1e12
        ref_tflop = world * B * n_fwd * flops_forward(N_FRAMES, hoisted=False) / 1e12
        roof = gemm_roofline(model, B)
        rec = {
            "metric": "mel_frames_per_sec", "value": value, "unit": "mel-frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": {"bf16": "bf16", "bf16x3": "bf16x3 (split-bf16 operands, fp32-class)",
                      "mxfp8": "mxfp8 (OCP e4m3 + E8M0 block scales for the four per-block GEMMs; attention and the rest bf16)"}[args.precision],
            "data": "synthetic (seeded random-init 335M weights, white-noise reference audio, random token ids)",
            "config": {"workload": f"F5-TTS 335M, {args.ode_points}-point {args.method} (={n_fwd} DiT forwards, CFG), "
                                   f"batch {B}/GPU x 10 s (N=937) utterances, hipGraph={not args.no_graph}"
                                   + (", + Vocos vocoder (mel -> waveform) in the timed region" if args.vocoder else ""),
                       "global_batch": world * B, "seq_len": N_FRAMES, "parallelism": f"dp{world} (utterance sharding)"},
            "rtf": world * B * 10.0 / (ms_per_step * 1e-3),
            # the reference's own definition (generate.py:183-189): seconds of GENERATED audio (reference trimmed off) per second
            "rtf_generated_only": world * B * (N_FRAMES - REF_SAMPLES // 256) * 256 / 24000.0 / (ms_per_step * 1e-3),
            "per_gpu_value": value / world,
            "executed_tflop_per_step": exec_tflop, "reference_tflop_per_step": ref_tflop,
            "whole_path_tflops": exec_tflop / (ms_per_step * 1e-3),
            "whole_path_frac_of_bf16_peak": exec_tflop / (ms_per_step * 1e-3) / (world * BF16_PEAK_TFLOPS),
            "weights_load_s": load_s, "weights_broadcast_ms": bcast_ms,
            "roofline": roof,
        }
        if world == 1 and not args.no_cpu_baseline:
            rec["cpu_baseline"] = cpu_baseline(weights)
        print(json.dumps(rec))
    if dist_on:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
