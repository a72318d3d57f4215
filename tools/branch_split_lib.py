import os, sys
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

