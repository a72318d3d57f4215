"""Run the large-grid attention op a few times with one kernel version (argv[1]) for rocprofv3 --pmc (tools/gpu_attn_pmc.sh)."""
import ctypes as C, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E
lib, dev, P = E.load_library(), torch.device("cuda:0"), E.ptr
ver, prec = int(sys.argv[1]), (sys.argv[2] if len(sys.argv) > 2 else "f16")
D, H, N, nb = 1024, 16, 937, 64
npad = (N + 63) // 64 * 64
dt = E.operand_dtype(prec)
g = torch.Generator(device="cpu").manual_seed(0)
qk = torch.randn(nb * N, 2 * D, generator=g).to(dev).to(dt)
vt = torch.zeros(nb * H, 64, npad, dtype=dt, device=dev)
vt[..., :N] = torch.randn(nb * H, 64, N, generator=g).to(dev).to(dt)
ao = torch.empty(nb * N, D, dtype=dt, device=dev)
E.check(lib.f5_debug_set_attn_version(ver))
with E.operand_type(prec):
    for _ in range(4):
        E.check(lib.f5_op_attention(P(qk), P(None), P(vt), P(None), P(ao), P(None), P(None), nb, H, N, npad, D, C.c_float(0.125), 0, E.stream_ptr(dev)))
torch.cuda.synchronize()
