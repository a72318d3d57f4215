"""Per-kernel in-graph time vs batch (nb x 937 rows): T = F + W*nb.  Which kernels carry the fixed cost at B=1?"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools import microbench as mb
import ctypes as C, torch
E, lib, P, dev = mb.E, mb.lib, mb.P, mb.dev
from tools.branch_split_lib import Chain, D, FF, H, N, npad

def graph_time(fn, reps=44, iters=10):
    side = torch.cuda.Stream()
    fn(E.stream_ptr(dev)); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(reps):
            fn(E.stream_ptr(dev))
    for _ in range(3):
        g.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        g.replay()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters / reps * 1e3

def main():
  res = {}
  for nb in (1, 2, 4, 8):
      c = Chain(nb)
      ops = dict(
          ln=lambda st: E.check(lib.f5_op_ln_modulate(P(c.x), P(c.sc), P(c.sh), P(c.h), P(None), c.M, D, st)),
          qkv=lambda st: E.check(lib.f5_op_gemm(P(c.h), P(None), P(c.wqkv), P(None), P(c.bq), P(None), P(c.qkv), P(None), c.M, 3 * D, D, D, D, 3 * D, 1, 1, st)),
          attn=lambda st: E.check(lib.f5_op_attention(P(c.qk), P(None), P(c.vt), P(None), P(c.ao), P(None), P(None), c.nb, H, N, npad, D, C.c_float(0.125), 0, st)),
          oproj=lambda st: E.check(lib.f5_op_gemm_resid_gate(P(c.ao), P(None), P(c.wo), P(None), P(c.bo), P(c.gate), P(None), P(c.x), c.M, D, D, D, D, D, 1, st)),
          ff1=lambda st: E.check(lib.f5_op_gemm(P(c.h), P(None), P(c.w1), P(None), P(c.b1), P(None), P(c.ff), P(None), c.M, FF, D, D, D, FF, 1, 2, st)),
          ff2=lambda st: E.check(lib.f5_op_gemm_resid_gate(P(c.ff), P(None), P(c.w2), P(None), P(c.bo), P(c.gate), P(None), P(c.x), c.M, D, FF, FF, FF, D, 1, st)),
      )
      for k, fn in ops.items():
          res.setdefault(k, []).append(round(graph_time(fn), 2))
  for k, v in res.items():
      print(json.dumps({"op": k, "us_nb1_2_4_8": v}))

if __name__ == "__main__":
    main()
