"""Round-3 probe 7: the role-split attention kernel (f5_attn2r, wide = 2) against the shipped large-grid kernel (f5_attn2f, wide = 1):
results (max difference, against an fp64 reference on a subset), repeat runs, timing at 64 x 16 heads x 937 (f16, q pre-multiplied)."""
import ctypes as C, json, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E
from tools.r3_probe2 import ev_time, lib, dev, P, st
opd = torch.float16
H, D = 16, 1024
with E.operand_type("f16"):
    for premul in (1, 0):
        lib.f5_debug_set_op_q_premul(C.c_float(0.125 * 1.4426950408889634 if premul else 0.0))
        for B, N, lens in ((64, 937, None), (3, 600, [600, 433, 65]), (2, 1100, [1100, 1099]), (2, 937, None)):
            npad = (N + 63) // 64 * 64
            g = torch.Generator(device="cpu").manual_seed(B * N)
            qk = (torch.randn(B * N, 2 * D, generator=g) * 0.6).to(dev).to(opd)
            vt = torch.randn(B * H, 64, npad, generator=g).to(dev).to(opd)
            vt[:, :, N:] = 0
            kv = torch.tensor(lens, dtype=torch.int32, device=dev) if lens else None
            outs = {}
            rec = dict(premul=premul, B=B, N=N, ragged=bool(lens))
            for wide in (1, 2, -1):
                E.check(lib.f5_debug_set_attn_wide(wide))
                E.check(lib.f5_debug_set_attn_kvsplit(1 if wide >= 0 else -1))
                ao = torch.zeros(B * N, D, dtype=opd, device=dev)
                fn = lambda: E.check(lib.f5_op_attention(P(qk), P(None), P(vt), P(None), P(ao), P(None), P(kv), B, H, N, npad, D, C.c_float(0.125), 0, st()))
                fn(); torch.cuda.synchronize()
                outs[wide] = ao.clone()
                rep = True
                for _ in range(5):
                    ao.zero_(); fn(); torch.cuda.synchronize()
                    rep = rep and torch.equal(ao, outs[wide])
                us = ev_time(fn, iters=20, warm=3)
                rec[f"wide{wide}"] = dict(us=round(us, 1), tf=round(4.0 * B * H * N * N * 64 / us / 1e6), repeatable=rep)
            d = (outs[1].float() - outs[2].float()).abs()
            rec["max_diff_2r_vs_2f"] = float(d.max()); rec["mean_abs_out"] = float(outs[1].float().abs().mean())
            # fp64 reference for batch element 0, head 3
            b0, h0 = 0, 3
            n0 = lens[b0] if lens else N
            q = qk[b0 * N:b0 * N + N, h0 * 64:(h0 + 1) * 64].double() * (1.0 if not premul else 1.0)
            k = qk[b0 * N:b0 * N + n0, D + h0 * 64:D + (h0 + 1) * 64].double()
            v = vt[b0 * H + h0, :, :n0].double().T
            sc = q @ k.T * (0.6931471805599453 if premul else 0.125)     # exp2 units -> natural log units when q carries log2(e) * scale
            ref = torch.softmax(sc, dim=-1) @ v
            got = outs[2][b0 * N:b0 * N + N, h0 * 64:(h0 + 1) * 64].double()
            rec["max_err_vs_fp64"] = float((got - ref).abs().max())
            print(json.dumps(rec), flush=True)
    E.check(lib.f5_debug_set_attn_wide(-1)); E.check(lib.f5_debug_set_attn_kvsplit(-1))
    lib.f5_debug_set_op_q_premul(C.c_float(0.0))
