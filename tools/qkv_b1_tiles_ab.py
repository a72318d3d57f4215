"""QKV projection (+ bias, RoPE, head split) at the batch-1 shape (M = 1874), f16 operands, q carrying scale * log2(e), same call,
interleaved rounds: auto tiles (64x128 register-staged, straight) vs the 8-wave 128x256 rings (tiles 12 / 13) with straight or
transposed q / k wave tiles.  usage: python tools/qkv_b1_tiles_ab.py [f16|bf16]"""
import ctypes as C
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402
from tools.attn_prio_bench import graph_time  # noqa: E402

lib, dev, P = E.load_library(), torch.device("cuda:0"), E.ptr
D, H, N, nb = 1024, 16, 937, 2
npad, M = 960, 2 * 937
QPRE = 0.125 * 1.4426950408889634


def main(prec):
    dt = E.operand_dtype(prec)
    g = torch.Generator(device="cpu").manual_seed(0)
    st0 = E.stream_ptr(dev)
    cos_t, sin_t = torch.empty((N, 32), device=dev), torch.empty((N, 32), device=dev)
    tt = [torch.empty((32, N), device=dev) for _ in range(4)]
    E.check(lib.f5_op_rope_table(P(cos_t), P(sin_t), N, 64, st0))
    E.check(lib.f5_op_rope_table_t(P(tt[0]), P(tt[1]), P(tt[2]), P(tt[3]), N, 64, C.c_float(QPRE), st0))
    E.check(lib.f5_debug_set_op_q_premul(C.c_float(QPRE)))
    x = torch.randn(M, D, generator=g).to(dev).to(dt)
    w = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).to(dt)
    bias = torch.zeros(3 * D, device=dev)
    qk = torch.empty(M, 2 * D, dtype=dt, device=dev)
    vt = torch.zeros(nb * H, 64, npad, dtype=dt, device=dev)
    fn = lambda st: E.check(lib.f5_op_qkv_rope(P(x), P(None), P(w), P(None), P(bias), P(cos_t), P(sin_t), P(qk), P(None), P(vt), P(None),
                                               nb, N, npad, H, D, 1, st))
    res, outs = {}, {}
    with E.operand_type(prec):
        for rnd in range(4):
            for tile in (0, 12, 13):
                for tr in ((False,) if tile == 0 else (False, True)):
                    E.check(lib.f5_debug_set_gemm_tile(tile))
                    E.check(lib.f5_debug_set_op_rope_tables_t(*([P(t) for t in tt] if tr else [P(None)] * 4)))
                    k = f"tile{tile}.{'transposed' if tr else 'straight'}"
                    res.setdefault(k, []).append(round(graph_time(fn, reps=44), 1))
                    outs[k] = qk.clone()
    E.check(lib.f5_debug_set_gemm_tile(0))
    E.check(lib.f5_debug_set_op_rope_tables_t(P(None), P(None), P(None), P(None)))
    E.check(lib.f5_debug_set_op_q_premul(C.c_float(0.0)))
    print(json.dumps(dict(op="qkv", M=M, prec=prec, us=res, identical_to_auto={k: bool(torch.equal(v, outs["tile0.straight"])) for k, v in outs.items()})),
          flush=True)


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "f16")
