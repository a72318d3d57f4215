"""Does running the cond / null CFG branches as two concurrent graph branches (M=937 each) beat one batched chain (M=1874)?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
import ctypes as C, torch
E, lib, P, dev = mb.E, mb.lib, mb.P, mb.dev
D, FF, H, N = 1024, 2048, 16, 937
npad = (N + 63) // 64 * 64

class Chain:
    def __init__(self, nb):
        M = nb * N
        self.nb, self.M = nb, M
        r = mb.rnd
        self.x = torch.zeros(M, D, device=dev)
        self.sc, self.sh = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
        self.h = r(M, D)
        self.wqkv, self.wo, self.w1, self.w2 = r(3 * D, D), r(D, D), r(FF, D), r(D, FF)
        self.bq, self.bo, self.b1 = torch.zeros(3 * D, device=dev), torch.zeros(D, device=dev), torch.zeros(FF, device=dev)
        self.gate = torch.ones(D, device=dev) * 0.01
        self.qk = r(M, 2 * D)
        self.qkv = r(M, 3 * D)
        self.vt = r(nb * H, 64, npad)
        self.ao = r(M, D)
        self.ff = r(M, FF)
    def block(self, st):
        c = self
        E.check(lib.f5_op_ln_modulate(P(c.x), P(c.sc), P(c.sh), P(c.h), P(None), c.M, D, st))
        E.check(lib.f5_op_gemm(P(c.h), P(None), P(c.wqkv), P(None), P(c.bq), P(None), P(c.qkv), P(None), c.M, 3 * D, D, D, D, 3 * D, 1, 1, st))
        E.check(lib.f5_op_attention(P(c.qk), P(None), P(c.vt), P(None), P(c.ao), P(None), P(None), c.nb, H, N, npad, D, C.c_float(0.125), 0, st))
        E.check(lib.f5_op_gemm_resid_gate(P(c.ao), P(None), P(c.wo), P(None), P(c.bo), P(c.gate), P(None), P(c.x), c.M, D, D, D, D, D, 1, st))
        E.check(lib.f5_op_ln_modulate(P(c.x), P(c.sc), P(c.sh), P(c.h), P(None), c.M, D, st))
        E.check(lib.f5_op_gemm(P(c.h), P(None), P(c.w1), P(None), P(c.b1), P(None), P(c.ff), P(None), c.M, FF, D, D, D, FF, 1, 2, st))
        E.check(lib.f5_op_gemm_resid_gate(P(c.ff), P(None), P(c.w2), P(None), P(c.bo), P(c.gate), P(None), P(c.x), c.M, D, FF, FF, FF, D, 1, st))

def run(graph, iters=10):
    for _ in range(3):
        graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        graph.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters

L = 22
side = torch.cuda.Stream()
one = Chain(2)
a, b = Chain(1), Chain(1)
# warm
for c in (one, a, b):
    c.block(E.stream_ptr(dev))
torch.cuda.synchronize()
g1 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g1, stream=side):
    for _ in range(L):
        one.block(E.stream_ptr(dev))
g2 = torch.cuda.CUDAGraph()
s2 = torch.cuda.Stream()
with torch.cuda.graph(g2, stream=side):
    cur = torch.cuda.current_stream()
    s2.wait_stream(cur)
    for _ in range(L):
        a.block(C.c_void_p(cur.cuda_stream))
    with torch.cuda.stream(s2):
        for _ in range(L):
            b.block(C.c_void_p(s2.cuda_stream))
    cur.wait_stream(s2)
g3 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g3, stream=side):
    for _ in range(L):
        a.block(E.stream_ptr(dev))
t1, t2, t3 = run(g1), run(g2), run(g3)
print(json.dumps(dict(batched_M1874_ms=round(t1, 3), two_branches_M937_ms=round(t2, 3), single_M937_ms=round(t3, 3))))
