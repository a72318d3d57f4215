"""Round-3 probe 11: hazard (a) of the round-2 verdict -- the LDS-free QKV epilogue (tools/experiments/qkv_direct_epilogue.patch
on commit db15be7, library passed in F5_PROBE_LIB) returned a few hundred wrong elements on the 4-wave kernels, 'on lanes 48-63
only and differently on every launch'.  Repeat launches per tile selection, count elements that differ from the LDS-staged
epilogue (same kernel family, straight tiles) and between repeats, and say where they sit (row within the 32-row MFMA block, column
within the 8-feature group: in the transposed layout lane = row % 32 + 32 * ((col % 8) / 4))."""
import ctypes as C, json, os, sys
from pathlib import Path
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E
E._LIB_PATH = Path(os.environ["F5_PROBE_LIB"]).resolve()
lib = E.load_library()
dev = torch.device("cuda:0")
P = E.ptr
st = lambda: E.stream_ptr(dev)
tiles = [int(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "2", "5", "12"])]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
B, H, N = 2, 16, 937
D = H * 64
npad = (N + 63) // 64 * 64
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
for tile in tiles:
    E.check(lib.f5_debug_set_gemm_tile(tile))
    qk0 = torch.zeros(B * N, 2 * D, dtype=opd, device=dev); vt0 = torch.zeros(B * H, 64, npad, dtype=opd, device=dev)
    E.check(lib.f5_op_qkv_rope(P(x), P(None), P(w), P(None), P(bias), P(cos_t), P(sin_t), P(qk0), P(None), P(vt0), P(None), B, N, npad, H, D, 1, st()))
    torch.cuda.synchronize()
    bad_runs, hist_lane, nbad_total, first = 0, {}, 0, None
    run0, differs_from_run0 = None, 0
    for r in range(reps):
        qk = torch.full((B * N, 2 * D), 7.0, dtype=opd, device=dev); vt = torch.zeros(B * H, 64, npad, dtype=opd, device=dev)
        E.check(lib.f5_op_qkv_rope_direct(P(x), P(None), P(w), P(None), P(bias), P(tt[0]), P(tt[1]), P(tt[2]), P(tt[3]), P(qk), P(None), P(vt), P(None), B, N, npad, H, D, 1, st()))
        torch.cuda.synchronize()
        if run0 is None: run0 = qk.clone()
        elif not torch.equal(run0, qk): differs_from_run0 += 1
        d = (qk.float() - qk0.float()).abs()
        tol = 2.0 ** -8 * qk0.float().abs().clamp(min=1.0)          # (transposed tiles may contract differently: 1 ulp)
        badmask = d > tol
        nb = int(badmask.sum())
        vbad = int((vt != vt0).sum())
        if nb or vbad:
            bad_runs += 1; nbad_total += nb
            idx = badmask.nonzero()
            for rr, cc in idx[:4000].tolist():
                lane = (rr % 32) + 32 * ((cc % 8) // 4)             # (row tiles start at multiples of 32 for every tile size used here)
                hist_lane[lane] = hist_lane.get(lane, 0) + 1
            if first is None:
                first = dict(run=r, n_bad_qk=nb, n_bad_vt=vbad, sample=[(int(a), int(b), float(qk[a, b]), float(qk0[a, b])) for a, b in idx[:6].tolist()])
    lanes = sorted(hist_lane)
    print(json.dumps(dict(lib=os.path.basename(str(E._LIB_PATH)), tile=tile, reps=reps, runs_that_differ_from_run0=differs_from_run0, runs_with_wrong_elements=bad_runs, wrong_elements_total=nbad_total,
                          lanes=[lanes[0], lanes[-1]] if lanes else None, lanes_below_48=sum(v for k, v in hist_lane.items() if k < 48), first=first)), flush=True)
E.check(lib.f5_debug_set_gemm_tile(0))
