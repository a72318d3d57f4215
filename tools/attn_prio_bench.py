"""Large-grid attention kernel (B=32: 64 x 16 heads x N=937): issue-priority scheme A/B, interleaved rounds in one process,
both operand builds.  prio 0 = s_setprio 1 around the MFMA clusters (round-1 kernel), 1 = no priority changes, 2 = the softmax
VALU section holds priority and the MFMA clusters run at priority 0."""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402

lib, dev, P = E.load_library(), torch.device("cuda:0"), E.ptr
D, H, N = 1024, 16, 937
npad = (N + 63) // 64 * 64


def graph_time(fn, reps=12, iters=5):
    side = torch.cuda.Stream()
    fn(E.stream_ptr(dev)); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            fn(E.stream_ptr(dev))
    for _ in range(2):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / reps * 1e3


def main():
    nbs = [int(x) for x in (sys.argv[1:] or ["64"])]
    for nb in nbs:
        for prec in ("bf16", "f16"):
            dt = E.operand_dtype(prec)
            g = torch.Generator(device="cpu").manual_seed(0)
            qk = torch.randn(nb * N, 2 * D, generator=g).to(dev).to(dt)
            vt = torch.zeros(nb * H, 64, npad, dtype=dt, device=dev)
            vt[..., :N] = torch.randn(nb * H, 64, N, generator=g).to(dev).to(dt)
            ao = torch.empty(nb * N, D, dtype=dt, device=dev)
            fn = lambda st: E.check(lib.f5_op_attention(P(qk), P(None), P(vt), P(None), P(ao), P(None), P(None), nb, H, N, npad, D,
                                                        C.c_float(0.125), 0, st))
            # (kernel version, variant bits): 2 = round-1 kernels; 5 / 6 = in-wave pipelined kernel with 8 / 4 waves per workgroup;
            # variant 1 = single-issue softmax VALU, 2 = one workgroup per CU (4 waves: one wave per SIMD)
            arms = [(2, 0), (2, 8)]     # variant 8 = eager O rescale (A/B of the lazy one), 4 = plain 2-D block numbering
            res = {f"{v}.{b}": [] for v, b in arms}
            with E.operand_type(prec):
                for rnd in range(4):
                    for ver, bits in arms:
                        E.check(lib.f5_debug_set_attn_version(ver))
                        E.check(lib.f5_debug_set_attn_variant(bits))
                        res[f"{ver}.{bits}"].append(graph_time(fn))
            E.check(lib.f5_debug_set_attn_version(2))
            E.check(lib.f5_debug_set_attn_variant(0))
            fl = 4.0 * nb * H * N * N * 64
            print(json.dumps(dict(nb=nb, prec=prec, us={k: [round(x, 1) for x in v] for k, v in res.items()},
                                  tflops_best={k: round(fl / min(v) / 1e6) for k, v in res.items()})), flush=True)


if __name__ == "__main__":
    main()
