"""Round-5 probe (GPU box, measurement only): where does the time between two main loops of the 256x256 GEMM go, and is it a
property of ONE CU (its own latency chain) or of the CHIP (256 CUs bursting their epilogues at the same moment)?

Needs the measurement build (F5_PROBE=1 bash f5_tts_mlx_amd/csrc/build.sh -> libf5tts_hip_probe.so), which this script loads through
F5TTS_HIP_LIB; the product library ignores the switches (compile-time false).  Per block GEMM shape at M = 59 968, fp16, workload-like
operands, interleaved rounds, min of the rounds in us:

  vendor            torch.matmul (hipBLASLt)
  full              the product launch
  main_loop         debug flag 1: K loop only
  no_store          0x10000: epilogue runs (arithmetic, LDS staging), global stores dropped
  no_math           0x20000: no GELU / rotation arithmetic (no table loads)
  no_store_no_math  both: what the staging itself costs
  nt                0x40000 (residual GEMMs): x read / written non-temporally
  shift_<mode>_<us> first-round workgroups phase-shifted by <us> (mode 0: every other CU of an XCD, 1: every other XCD, 2: four phases)
  nband_<n>         (QKV) band width of the tile numbering

usage: F5TTS_HIP_LIB=f5_tts_mlx_amd/csrc/libf5tts_hip_probe.so python tools/r5_epilogue_probe.py > gpurun_out/.../epilogue_probe.jsonl
"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402

lib = E.load_library()
dev = torch.device("cuda:0")
P = E.ptr
st = lambda: E.stream_ptr(dev)   # noqa: E731
M_ROWS, D, FF, H, N_FRAMES = 59968, 1024, 2048, 16, 937


def ev_time(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    assert "probe" in str(E.library_path()), f"load the measurement build through F5TTS_HIP_LIB (got {E.library_path()})"
    opd = torch.float16
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda std, *s: (torch.randn(*s, generator=g) * std).to(dev).to(opd)   # noqa: E731
    shapes = [("qkv", 3 * D, D, "qkv"), ("out_proj", D, D, "resid"), ("ff1", FF, D, "gelu"), ("ff2", D, FF, "resid")]
    only = sys.argv[1:]
    with E.operand_type("f16"):
        for name, N, K, kind in shapes:
            if only and name not in only:
                continue
            a, w = mk(1.0, M_ROWS, K), mk(K ** -0.5, N, K)
            bias = torch.zeros(N, device=dev)
            gate = torch.full((N,), 0.5, device=dev)
            xres = torch.zeros(M_ROWS, D, device=dev)
            out16 = torch.empty(M_ROWS, N, dtype=opd, device=dev)
            npad = (N_FRAMES + 63) // 64 * 64
            cos_t, sin_t = torch.ones(N_FRAMES, 32, device=dev), torch.zeros(N_FRAMES, 32, device=dev)
            tt = [torch.empty(64 * N_FRAMES, device=dev) for _ in range(2)]
            E.check(lib.f5_op_rope_table(P(cos_t), P(sin_t), N_FRAMES, 64, st()))
            E.check(lib.f5_op_rope_table_g4(P(tt[0]), P(tt[1]), N_FRAMES, 64, E.C.c_float(1.0), st()))
            qk = torch.empty(M_ROWS, 2 * D, dtype=opd, device=dev)
            vt = torch.zeros(64 * H, 64, npad, dtype=opd, device=dev)
            wt = w.t()
            vend = lambda: torch.matmul(a, wt, out=out16)          # noqa: E731
            if kind == "resid":
                ours = lambda: E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bias), P(gate), P(None), P(xres),   # noqa: E731
                                                                 M_ROWS, D, K, K, K, D, 1, st()))
            elif kind == "gelu":
                ours = lambda: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(out16), P(None), M_ROWS, N, K, K, K, N, 1, 2, st()))   # noqa: E731
            else:
                E.check(lib.f5_debug_set_op_rope_tables_g4(P(tt[0]), P(tt[1])))     # the transposed q / k tiles, as sample()
                ours = lambda: E.check(lib.f5_op_qkv_rope(P(a), P(None), P(w), P(None), P(bias), P(cos_t), P(sin_t), P(qk), P(None), P(vt),   # noqa: E731
                                                          P(None), 64, N_FRAMES, npad, H, D, 1, st()))
            variants = [("vendor", vend, 0, None), ("full", ours, 0, None), ("main_loop", ours, 1, None), ("no_store", ours, 0x10000, None),
                        ("no_math", ours, 0x20000, None), ("no_store_no_math", ours, 0x30000, None)]
            if kind == "resid":
                variants.append(("nt", ours, 0x40000, None))
            for mode in (0, 1, 2):
                for us in (4, 8, 14, 20):
                    variants.append((f"shift_m{mode}_{us}us", ours, (mode << 28) | ((2 * us) << 20), None))
            if kind == "qkv":
                for nb in (0, 2, 3, 6, 12):
                    variants.append((f"nband_{nb}", ours, 0, nb))
            rec = dict(kind="epilogue_probe", shape=name, M=M_ROWS, N=N, K=K, us={})
            flops = 2.0 * M_ROWS * N * K
            for rnd in range(3):
                for key, fn, flags, nband in variants:
                    lib.f5_debug_set_gemm_flags(flags)
                    if nband is not None:
                        lib.f5_debug_set_gemm_nband(nband)
                    rec["us"].setdefault(key, []).append(round(ev_time(fn), 1))
                    if nband is not None:
                        lib.f5_debug_set_gemm_nband(4)
                lib.f5_debug_set_gemm_flags(0)
            rec["min_us"] = {k: min(v) for k, v in rec["us"].items()}
            rec["tflops"] = {k: round(flops / min(v) / 1e6) for k, v in rec["us"].items()}
            ml = rec["min_us"]["main_loop"]
            rec["over_main_loop_us"] = {k: round(v - ml, 1) for k, v in rec["min_us"].items()}
            if kind == "qkv":
                E.check(lib.f5_debug_set_op_rope_tables_g4(P(None), P(None)))
            print(json.dumps(rec), flush=True)


if __name__ == "__main__":
    main()
