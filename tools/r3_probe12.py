"""Round-3 probe 12: where the wrong elements of hazard (a) sit (one failing run of tile 2 = 64x128 register-staged, 4 waves)."""
import ctypes as C, json, os, sys, collections
from pathlib import Path
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E
E._LIB_PATH = Path(os.environ["F5_PROBE_LIB"]).resolve()
lib = E.load_library()
dev = torch.device("cuda:0")
P = E.ptr
st = lambda: E.stream_ptr(dev)
tile = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, H, N = 2, 16, 937
D = H * 64
npad = 960
opd = torch.float16
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn(B * N, D, generator=g).to(dev).to(opd)
w = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).to(opd)
bias = (torch.randn(3 * D, generator=g) * 0.1).to(dev)
E.check(lib.f5_op_set_operand_type(1))
cos_t, sin_t = torch.empty(N, 32, device=dev), torch.empty(N, 32, device=dev)
E.check(lib.f5_op_rope_table(P(cos_t), P(sin_t), N, 64, st()))
tt = [torch.empty(32, N, device=dev) for _ in range(4)]
E.check(lib.f5_op_rope_table_t(P(tt[0]), P(tt[1]), P(tt[2]), P(tt[3]), N, 64, C.c_float(1.0), st()))
E.check(lib.f5_debug_set_gemm_tile(tile))
qk0 = torch.zeros(B * N, 2 * D, dtype=opd, device=dev); vt0 = torch.zeros(B * H, 64, npad, dtype=opd, device=dev)
E.check(lib.f5_op_qkv_rope(P(x), P(None), P(w), P(None), P(bias), P(cos_t), P(sin_t), P(qk0), P(None), P(vt0), P(None), B, N, npad, H, D, 1, st()))
torch.cuda.synchronize()
# fp64 reference pieces: un-rotated projection a = x w^T + b for q | k
a_ref = (x.double() @ w[:2 * D].double().T + bias[:2 * D].double())
for r in range(3):
    qk = torch.full((B * N, 2 * D), 7.0, dtype=opd, device=dev); vt = torch.zeros(B * H, 64, npad, dtype=opd, device=dev)
    E.check(lib.f5_op_qkv_rope_direct(P(x), P(None), P(w), P(None), P(bias), P(tt[0]), P(tt[1]), P(tt[2]), P(tt[3]), P(qk), P(None), P(vt), P(None), B, N, npad, H, D, 1, st()))
    torch.cuda.synchronize()
    d = (qk.float() - qk0.float()).abs()
    bad = (d > 2.0 ** -8 * qk0.float().abs().clamp(min=1.0)).nonzero().tolist()
    by_tile = collections.Counter((rr // 64, cc // 128) for rr, cc in bad)            # workgroup tile (64 x 128)
    by_wave = collections.Counter(((rr % 64) // 32, (cc % 128) // 64) for rr, cc in bad)   # wave inside the workgroup (2 x 2 waves of 32 x 64)
    by_reg = collections.Counter(((cc % 64) // 32, (cc % 32) // 8, cc % 4) for rr, cc in bad)   # (nb, rg, element of the lane's 4)
    by_rowq = collections.Counter((rr % 32) // 16 for rr, cc in bad)
    by_hi = collections.Counter((cc % 8) // 4 for rr, cc in bad)
    # is the wrong value the value of ANOTHER element?  compare with the reference rotated with the position of a different row
    rec = dict(run=r, n_bad=len(bad), workgroup_tiles=len(by_tile), max_per_tile=max(by_tile.values()) if by_tile else 0,
               wave_in_wg=dict((str(k), v) for k, v in by_wave.items()), nb_rg_e=dict((str(k), v) for k, v in sorted(by_reg.items())),
               row_half=dict(by_rowq), hi=dict(by_hi))
    # for a few bad elements: got, expected, and what (acc + bias) would be WITHOUT the rotation; and the pair partner's expected value
    ex = []
    for rr, cc in bad[:8]:
        partner = cc ^ 1
        ex.append(dict(row=rr, col=cc, got=float(qk[rr, cc]), want=float(qk0[rr, cc]), unrotated=round(float(a_ref[rr, cc]), 4),
                       partner_want=float(qk0[rr, partner]), partner_got=float(qk[rr, partner]), partner_unrot=round(float(a_ref[rr, partner]), 4)))
    rec["examples"] = ex
    print(json.dumps(rec), flush=True)
E.check(lib.f5_debug_set_gemm_tile(0))
