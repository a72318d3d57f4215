"""Round-4 probe: timing-only ablations of the in-wave pipelined attention kernel (f5_debug_set_attn_pipe 11..16) at 64 x 16 x 937."""
import ctypes as C, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E
from tools.yardstick import ev_time, lib, dev, P, st, D, H, N_FRAMES
opd = torch.float16
B = 64
npad = (N_FRAMES + 63) // 64 * 64
g = torch.Generator(device="cpu").manual_seed(1)
names = {0: "v2f", 1: "v2p", 11: "v2p no exp (mov instead)", 12: "v2p no softmax VALU", 13: "v2p no LDS fragment reads in the loop",
         14: "v2p no waits / barriers / staging", 15: "v2p no MFMAs", 16: "v2p no s_nop guards, no slow-path check", 17: "v2p guards, no slow-path check", 18: "v2p slow-path check, no guards"}
with E.operand_type("f16"):
    lib.f5_debug_set_op_q_premul(C.c_float(0.125 * 1.4426950408889634))
    qk = (torch.randn(B * N_FRAMES, 2 * D, generator=g) * 0.6).to(dev).to(opd)
    vt = torch.randn(B * H, 64, npad, generator=g).to(dev).to(opd)
    vt[:, :, N_FRAMES:] = 0
    ao = torch.zeros(B * N_FRAMES, D, dtype=opd, device=dev)
    fn = lambda: E.check(lib.f5_op_attention(P(qk), P(None), P(vt), P(None), P(ao), P(None), P(None), B, H, N_FRAMES, npad, D, C.c_float(0.125), 0, st()))
    res = {}
    for rnd in range(2):
        for v in names:
            E.check(lib.f5_debug_set_attn_pipe(v))
            res.setdefault(v, []).append(round(ev_time(fn, iters=10), 1))
    E.check(lib.f5_debug_set_attn_pipe(0))
    lib.f5_debug_set_op_q_premul(C.c_float(0.0))
for v, us in res.items():
    print(json.dumps(dict(variant=names[v], us=us)), flush=True)
