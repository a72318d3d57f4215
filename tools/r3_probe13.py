"""Round-3 probe 13: hazard (a) caught in the act -- library variant HZ=8 re-computes 'c a0 - s a1' with two un-packed inline-asm
instructions next to the compiler's packed group and records the operands when they disagree."""
import ctypes as C, json, os, sys
from pathlib import Path
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E
E._LIB_PATH = Path(os.environ["F5_PROBE_LIB"]).resolve()
lib = E.load_library()
dev = torch.device("cuda:0")
P = E.ptr
st = lambda: E.stream_ptr(dev)
B, H, N = 2, 16, 937
D = H * 64
npad = 960
opd = torch.float16
g = torch.Generator(device="cpu").manual_seed(5)
x = torch.randn(B * N, D, generator=g).to(dev).to(opd)
w = (torch.randn(3 * D, D, generator=g) * D ** -0.5).to(dev).to(opd)
bias = (torch.randn(3 * D, generator=g) * 0.1).to(dev)
E.check(lib.f5_op_set_operand_type(1))
tt = [torch.empty(32, N, device=dev) for _ in range(4)]
E.check(lib.f5_op_rope_table_t(P(tt[0]), P(tt[1]), P(tt[2]), P(tt[3]), N, 64, C.c_float(1.0), st()))
E.check(lib.f5_debug_set_gemm_tile(2))
buf = (C.c_float * (64 * 16))()
cnt = C.c_uint()
lib.f5_debug_read_hz(buf, C.byref(cnt))          # clear
for r in range(6):
    qk = torch.zeros(B * N, 2 * D, dtype=opd, device=dev); vt = torch.zeros(B * H, 64, npad, dtype=opd, device=dev)
    E.check(lib.f5_op_qkv_rope_direct(P(x), P(None), P(w), P(None), P(bias), P(tt[0]), P(tt[1]), P(tt[2]), P(tt[3]), P(qk), P(None), P(vt), P(None), B, N, npad, H, D, 1, st()))
    torch.cuda.synchronize()
    assert lib.f5_debug_read_hz(buf, C.byref(cnt)) == 0
    a = np.frombuffer(buf, dtype=np.float32).reshape(64, 16)
    print(f"run {r}: {cnt.value} disagreements")
    for k in range(min(cnt.value, 6)):
        row, c, lane, a0, a1, c0, s0, o0, chk, o1, rg, nb, t = a[k, :13]
        print(f"   row {int(row)} col {int(c)} lane {int(lane)} nb {int(nb)} rg {int(rg)}: a0 {a0:.5f} a1 {a1:.5f} c0 {c0:.5f} s0 {s0:.5f} | packed o0 {o0:.5f}  check {chk:.5f}  s0*a1 {t:.5f}  o0 + s0*a1 - c0*a0 = {o0 + t - c0 * a0:.6f}  | o1 {o1:.5f} (c0 a1 + s0 a0 = {c0 * a1 + s0 * a0:.5f})")
E.check(lib.f5_debug_set_gemm_tile(0))
