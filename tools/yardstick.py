"""Round-4 yardstick (GPU box, measurement only -- nothing here is on the product path).

VERDICT r3 item 2: the claim "1.2 PF is what a 16-bit MFMA main loop sustains on this workload's data" rested on this repo's own
kernels.  This tool puts independent numbers next to them, on the same box, in the same process, interleaved:

  gemm       the four DiT-block GEMM shapes (M = 59 968; N / K = 3072 / 1024, 1024 / 1024, 2048 / 1024, 1024 / 2048) in fp16:
             torch.matmul (hipBLASLt / rocBLAS, the vendor kernel) against f5_op_gemm (full launch with its fused epilogue, and
             main loop only = debug flag 1), each on workload-like operands (activations ~N(0,1), weights ~N(0,1/K)), on zeros and
             on constants -- the data-dependent clock is the point
  attention  64 x 16 heads x 937 x 64, fp16: torch scaled_dot_product_attention (the vendor flash kernel) against f5_op_attention
             (v2f and the in-wave pipelined v2p), workload-like and zero operands
  mfma       the pure-register MFMA loop (f5_op_mfma_peak) on constant and on workload-like operand registers: the rate
             bench.py prints as peak_measured_tflops
  clock      rocm-smi sclk / power sampled while each of the above loops (best effort; skipped if rocm-smi is missing)

usage: python tools/yardstick.py [gemm] [attention] [mfma] > profiles/r04/yardstick.jsonl
"""
import ctypes as C
import json
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402

lib = E.load_library()
dev = torch.device("cuda:0")
P = E.ptr
st = lambda: E.stream_ptr(dev)   # noqa: E731
M_ROWS, D, FF, H, N_FRAMES = 59968, 1024, 2048, 16, 937


def ev_time(fn, iters, warm=3):
    """average us of `iters` back-to-back launches (HIP events on the current stream)"""
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


class SmiSampler:
    """rocm-smi polled in a thread while a kernel loops: (sclk MHz, W) samples"""

    def __init__(self):
        self.samples, self._stop = [], False
        self.ok = subprocess.run(["which", "rocm-smi"], capture_output=True).returncode == 0

    def _run(self):
        while not self._stop:
            try:
                out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
                card = next(iter(json.loads(out).values()))
                sclk = next((v for k, v in card.items() if "sclk" in k.lower()), None)
                pw = next((v for k, v in card.items() if "power" in k.lower() and "(w)" in k.lower()), None)
                self.samples.append((sclk, pw))
            except Exception as e:   # noqa: BLE001
                self.samples.append(("error", str(e)[:60]))
            time.sleep(0.25)

    def measure(self, fn, seconds=2.0):
        if not self.ok:
            return None
        self.samples, self._stop = [], False
        th = threading.Thread(target=self._run)
        th.start()
        t0 = time.time()
        while time.time() - t0 < seconds:
            for _ in range(20):
                fn()
            torch.cuda.synchronize()
        self._stop = True
        th.join()
        return self.samples[1:][-4:]


def fills(M, K, N, opd, g):
    mk = lambda std, *s: (torch.randn(*s, generator=g) * std).to(dev).to(opd)   # noqa: E731
    return {
        "workload N(0,1) x N(0,1/K)": (mk(1.0, M, K), mk(K ** -0.5, N, K)),
        "zeros": (torch.zeros(M, K, dtype=opd, device=dev), torch.zeros(N, K, dtype=opd, device=dev)),
        "constant 0.5 / 0.03": (torch.full((M, K), 0.5, dtype=opd, device=dev), torch.full((N, K), 0.03, dtype=opd, device=dev)),
    }


def run_gemm(smi):
    opd = torch.float16
    g = torch.Generator(device="cpu").manual_seed(0)
    shapes = [("qkv", 3 * D, D, "qkv"), ("out_proj", D, D, "resid"), ("ff1", FF, D, "gelu"), ("ff2", D, FF, "resid")]
    with E.operand_type("f16"):
        for name, N, K, kind in shapes:
            bias = torch.zeros(N, device=dev)
            gate = torch.full((N,), 0.5, device=dev)
            xres = torch.zeros(M_ROWS, D, device=dev)
            out16 = torch.empty(M_ROWS, N, dtype=opd, device=dev)
            cos_t, sin_t = torch.ones(N_FRAMES, 32, device=dev), torch.zeros(N_FRAMES, 32, device=dev)
            npad = (N_FRAMES + 63) // 64 * 64
            qk = torch.empty(M_ROWS, 2 * D, dtype=opd, device=dev)
            vt = torch.zeros(64 * H, 64, npad, dtype=opd, device=dev)
            for fill, (a, w) in fills(M_ROWS, K, N, opd, g).items():
                wt = w.t()                                            # torch.matmul(a, w^T): the "NT" GEMM, W [N][K] K-contiguous like ours
                vend = lambda: torch.matmul(a, wt, out=out16)          # noqa: E731
                if kind == "resid":
                    ours = lambda: E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bias), P(gate), P(None), P(xres),   # noqa: E731
                                                                     M_ROWS, D, K, K, K, D, 1, st()))
                elif kind == "gelu":
                    ours = lambda: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(out16), P(None), M_ROWS, N, K, K, K, N, 1, 2, st()))   # noqa: E731
                else:
                    ours = lambda: E.check(lib.f5_op_qkv_rope(P(a), P(None), P(w), P(None), P(bias), P(cos_t), P(sin_t), P(qk), P(None), P(vt),   # noqa: E731
                                                              P(None), 64, N_FRAMES, npad, H, D, 1, st()))
                rec = dict(kind="gemm", shape=name, M=M_ROWS, N=N, K=K, fill=fill, us={}, tflops={})
                flops = 2.0 * M_ROWS * N * K
                for rnd in range(3):                                  # interleaved rounds: drift shows as spread, not as a kernel difference
                    for key, fn, flags in (("vendor_torch_matmul", vend, 0), ("f5_full", ours, 0), ("f5_main_loop_only", ours, 1)):
                        lib.f5_debug_set_gemm_flags(flags)
                        rec["us"].setdefault(key, []).append(round(ev_time(fn, iters=20), 1))
                    lib.f5_debug_set_gemm_flags(0)
                for key, v in rec["us"].items():
                    rec["tflops"][key] = round(flops / min(v) / 1e6)
                if fill.startswith("workload"):
                    rec["smi_vendor"] = smi.measure(vend)
                    rec["smi_f5_full"] = smi.measure(ours)
                print(json.dumps(rec), flush=True)


def run_attention(smi):
    opd = torch.float16
    B = 64
    npad = (N_FRAMES + 63) // 64 * 64
    g = torch.Generator(device="cpu").manual_seed(1)
    flops = 4.0 * B * H * N_FRAMES * N_FRAMES * 64
    with E.operand_type("f16"):
        lib.f5_debug_set_op_q_premul(C.c_float(0.125 * 1.4426950408889634))
        for fill in ("workload", "zeros"):
            if fill == "workload":
                qk = (torch.randn(B * N_FRAMES, 2 * D, generator=g) * 0.6).to(dev).to(opd)
                vt = torch.randn(B * H, 64, npad, generator=g).to(dev).to(opd)
                vt[:, :, N_FRAMES:] = 0
            else:
                qk = torch.zeros(B * N_FRAMES, 2 * D, dtype=opd, device=dev)
                vt = torch.zeros(B * H, 64, npad, dtype=opd, device=dev)
            ao = torch.zeros(B * N_FRAMES, D, dtype=opd, device=dev)
            ours = lambda: E.check(lib.f5_op_attention(P(qk), P(None), P(vt), P(None), P(ao), P(None), P(None), B, H, N_FRAMES, npad, D,   # noqa: E731
                                                       C.c_float(0.125), 0, st()))
            # the vendor kernel on the same values in its own layout: (B, H, N, 64) contiguous q / k / v
            q4 = qk[:, :D].reshape(B, N_FRAMES, H, 64).transpose(1, 2).contiguous()
            k4 = qk[:, D:].reshape(B, N_FRAMES, H, 64).transpose(1, 2).contiguous()
            v4 = vt[:, :, :N_FRAMES].reshape(B, H, 64, N_FRAMES).transpose(2, 3).contiguous()
            vend = lambda: torch.nn.functional.scaled_dot_product_attention(q4, k4, v4, scale=1.0)   # noqa: E731  (q carries the scale)
            rec = dict(kind="attention", shape=f"B={B} H={H} N={N_FRAMES} d=64", fill=fill, us={}, tflops={})
            for rnd in range(3):
                for key, fn, pipe in (("vendor_torch_sdpa", vend, 0), ("f5_v2f", ours, 0), ("f5_v2p", ours, 1)):
                    E.check(lib.f5_debug_set_attn_pipe(pipe))
                    rec["us"].setdefault(key, []).append(round(ev_time(fn, iters=20), 1))
                E.check(lib.f5_debug_set_attn_pipe(0))
            for key, v in rec["us"].items():
                rec["tflops"][key] = round(flops / min(v) / 1e6)
            if fill == "workload":
                # same operands -> the two f5 kernels must agree to a few 16-bit ulps
                E.check(lib.f5_debug_set_attn_pipe(0)); ours(); torch.cuda.synchronize(); ref = ao.float().clone()   # noqa: E702
                E.check(lib.f5_debug_set_attn_pipe(1)); ours(); torch.cuda.synchronize()                              # noqa: E702
                rec["v2p_vs_v2f_max_abs"] = float((ao.float() - ref).abs().max())
                rec["v2p_vs_v2f_mean_abs"] = float((ao.float() - ref).abs().mean())
                rec["out_mean_abs"] = float(ref.abs().mean())
                rec["smi_f5_v2p"] = smi.measure(ours)
                E.check(lib.f5_debug_set_attn_pipe(0))
                rec["smi_f5_v2f"] = smi.measure(ours)
                rec["smi_vendor"] = smi.measure(vend)
            print(json.dumps(rec), flush=True)
        lib.f5_debug_set_op_q_premul(C.c_float(0.0))


def mfma_peak(precision="f16", workload=True, blocks=1024, iters=4000):
    """TF/s of the pure-register MFMA loop; used by bench.py"""
    opd = E.operand_dtype(precision)
    ops = None
    if workload:
        g = torch.Generator(device="cpu").manual_seed(7)
        ops = torch.randn(16 * 64 * 8, generator=g).to(dev).to(opd)
    sink = torch.zeros(4, device=dev)
    fl = C.c_double()
    with E.operand_type(precision):
        fn = lambda: E.check(lib.f5_op_mfma_peak(P(ops), blocks, iters, P(sink), C.byref(fl), st()))   # noqa: E731
        us = min(ev_time(fn, iters=5, warm=2) for _ in range(3))
    return fl.value / us / 1e6, us


def run_mfma(smi):
    for prec in ("f16", "bf16"):
        for workload in (False, True):
            tf, us = mfma_peak(prec, workload)
            print(json.dumps(dict(kind="mfma_peak", precision=prec, operands="workload-like registers" if workload else "lane-constant registers",
                                  us=round(us, 1), tflops=round(tf))), flush=True)


if __name__ == "__main__":
    what = sys.argv[1:] or ["mfma", "gemm", "attention"]
    smi = SmiSampler()
    print(json.dumps(dict(kind="env", device=torch.cuda.get_device_name(0), torch=torch.__version__, rocm_smi=smi.ok)), flush=True)
    if "mfma" in what:
        run_mfma(smi)
    if "gemm" in what:
        run_gemm(smi)
    if "attention" in what:
        run_attention(smi)
