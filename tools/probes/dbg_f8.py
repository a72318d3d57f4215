import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from f5test import DEV, E, P, stream
from oracle import mx_oracle as MX
lib = E.load_library()
def q(x):
    rows, cols = x.shape
    xd = x.to(DEV).contiguous()
    qq = torch.empty((rows, cols), dtype=torch.uint8, device=DEV); sc = torch.empty((rows, cols // 32), dtype=torch.uint8, device=DEV)
    E.check(lib.f5_op_quantize_mx(P(xd), cols, P(qq), cols, P(sc), rows, cols, stream())); torch.cuda.synchronize()
    return qq, sc
def run(a, w, tag):
    M, K = a.shape; N = w.shape[0]
    a8, asc = q(a); w8, wsc = q(w)
    ref = MX.mx_dequantize(a8.cpu(), asc.cpu()) @ MX.mx_dequantize(w8.cpu(), wsc.cpu()).T
    out = torch.full((M, N), float('nan'), device=DEV)
    E.check(lib.f5_op_gemm_f8(P(a8), P(asc), P(w8), P(wsc), P(None), P(None), P(None), P(out), P(None), P(None), P(None), M, N, K, K, K, N, 0, stream()))
    torch.cuda.synchronize()
    d = (out.cpu().double() - ref).abs()
    print(tag, 'max err', float(d.max()), 'ref max', float(ref.abs().max()), 'asc uniq', asc.unique().tolist(), 'wsc uniq', wsc.unique().tolist())
    return out.cpu(), ref
g = torch.Generator().manual_seed(0)
M, N, K = 256, 256, 128
ones = torch.ones(M, K); 
run(ones, torch.ones(N, K), 'all ones')
a = torch.randint(-3, 4, (M, K), generator=g).float(); a[:, ::32] = 3.0   # every block has amax 3 -> same scale
w = torch.randint(-3, 4, (N, K), generator=g).float(); w[:, ::32] = 3.0
out, ref = run(a, w, 'ints, uniform scales')
a2 = a.clone(); a2[:, 32:64] *= 4.0        # K block 1 of A has a different scale
out, ref = run(a2, w, 'A block1 x4')
a3 = a.clone(); a3[7] *= 8.0               # one row of A scaled
out, ref = run(a3, w, 'A row7 x8')
print((out - ref).abs().amax(1)[:12])
w3 = w.clone(); w3[:, 96:128] *= 16.0
out, ref = run(a, w3, 'W block3 x16')
M, K = 256, 256
a = torch.randint(-3, 4, (M, K), generator=g).float(); a[:, ::32] = 3.0
w = torch.randint(-3, 4, (N, K), generator=g).float(); w[:, ::32] = 3.0
run(a, w, 'K=256 uniform')
