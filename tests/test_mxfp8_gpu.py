"""MX-fp8 path (BASELINE configs[4]): quantiser, 256x256x128 MX GEMM and its epilogues against the MX oracle."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from f5test import DEV, E, P, randn, report, rng, stream
from oracle import mx_oracle as MX

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def lib():
    return E.load_library()


def sync():
    torch.cuda.synchronize()


def _dev_quantize(lib, x):
    rows, cols = x.shape
    xd = x.to(DEV).contiguous()
    q = torch.empty((rows, cols), dtype=torch.uint8, device=DEV)
    sc = torch.empty((rows, cols // 32), dtype=torch.uint8, device=DEV)
    E.check(lib.f5_op_quantize_mx(P(xd), cols, P(q), cols, P(sc), rows, cols, stream()), "quantize_mx")
    sync()
    return q, sc


def test_quantize_mx_bit_exact(lib):
    r = rng(1)
    x = randn(r, 37, 256) * torch.logspace(-6, 3, 37)[:, None]          # 9 decades of magnitudes
    x[3, 32:64] = 0.0                                                    # an all-zero block
    x[5, 7] = 447.9
    x[6, 100] = -1e-30
    q, sc = _dev_quantize(lib, x)
    q_ref, sc_ref = MX.mx_quantize(x)
    assert torch.equal(sc.cpu(), sc_ref), "E8M0 scales must match the oracle bit for bit"
    assert torch.equal(q.cpu(), q_ref), "e4m3 bytes must match the oracle bit for bit"
    d = MX.mx_dequantize(q.cpu(), sc.cpu())
    blk = x.double().reshape(37, 8, 32)
    err = (d.reshape(37, 8, 32) - blk).abs().amax(-1)
    assert torch.all(err <= blk.abs().amax(-1) * 2.0 ** -3 + 1e-30)      # half a step at the top binade of (224, 448]


@pytest.mark.parametrize("M,N,K", [(256, 256, 128), (300, 512, 256), (1874, 1024, 1024), (700, 768, 2048)])
def test_gemm_f8_matches_dequantised_operands(lib, M, N, K):
    """exact e4m3 x e4m3 products, fp32 accumulation: against fp64 on the DEQUANTISED operands only summation order differs"""
    r = rng(M + N + K)
    a = randn(r, M, K) * torch.logspace(-2, 1, K // 32).repeat_interleave(32)[None, :]   # block scales differ along K
    w = randn(r, N, K, scale=K ** -0.5) * (1.0 + 3.0 * torch.rand((N, 1), generator=torch.Generator().manual_seed(1)))
    bias = randn(r, N, scale=0.1)
    a8, asc = _dev_quantize(lib, a)
    w8, wsc = _dev_quantize(lib, w)
    ad, wd = MX.mx_dequantize(a8.cpu(), asc.cpu()), MX.mx_dequantize(w8.cpu(), wsc.cpu())
    ref = ad @ wd.T + bias.double()
    bias_d = bias.to(DEV)
    out = torch.full((M, N), float("nan"), device=DEV)
    E.check(lib.f5_op_gemm_f8(P(a8), P(asc), P(w8), P(wsc), P(bias_d), P(None), P(None), P(out), P(None), P(None), P(None),
                              M, N, K, K, K, N, 0, stream()), "gemm_f8")
    sync()
    mx, _, _ = report(f"gemm mxfp8 {M}x{N}x{K} vs dequantised fp64", out.cpu(), ref)
    # the 64-term dot product inside one v_mfma_scale is not an exact fp32 chain (products are aligned to the largest
    # exponent before the add): a few 1e-5 of the output range
    assert mx <= 2e-4 * max(1.0, float(ref.abs().max()))
    full = a.double() @ w.double().T + bias.double()
    rel = float((out.cpu().double() - full).abs().mean() / full.abs().mean())
    print(f"   quantisation error vs fp64 of the unquantised operands: relative L1 {rel:.3e}")
    assert rel < 0.06
    # bf16 output
    ob = torch.zeros((M, N), dtype=torch.bfloat16, device=DEV)
    E.check(lib.f5_op_gemm_f8(P(a8), P(asc), P(w8), P(wsc), P(bias_d), P(None), P(None), P(None), P(ob), P(None), P(None),
                              M, N, K, K, K, N, 1, stream()), "gemm_f8 bf16")
    sync()
    assert float((ob.cpu().double() - ref).abs().max()) <= 1e-2 * max(1.0, float(ref.abs().max()))


def test_gemm_f8_epilogues(lib):
    r = rng(77)
    M, N, K = 650, 512, 256
    a, w, bias = randn(r, M, K), randn(r, N, K, scale=K ** -0.5), randn(r, N, scale=0.1)
    a8, asc = _dev_quantize(lib, a)
    w8, wsc = _dev_quantize(lib, w)
    ad, wd = MX.mx_dequantize(a8.cpu(), asc.cpu()), MX.mx_dequantize(w8.cpu(), wsc.cpu())
    y = ad @ wd.T + bias.double()
    bias_d = bias.to(DEV)
    # residual: x += gate * (y * keep)
    gate, x0 = randn(r, N), randn(r, M, N)
    keep = torch.from_numpy((r.random(M) > 0.3).astype(np.uint8))
    gate_d, keep_d, x = gate.to(DEV), keep.to(DEV), x0.to(DEV).clone()
    E.check(lib.f5_op_gemm_f8(P(a8), P(asc), P(w8), P(wsc), P(bias_d), P(gate_d), P(keep_d), P(x), P(None), P(None), P(None),
                              M, N, K, K, K, N, 4, stream()), "gemm_f8 resid")
    sync()
    ref = x0.double() + gate.double() * (y * keep.double()[:, None])
    assert float((x.cpu().double() - ref).abs().max()) <= 2e-4 * max(1.0, float(ref.abs().max()))
    # GELU -> MX-fp8 output: scales and bytes are the oracle's quantisation of the kernel's own fp32 GELU values, so compare
    # dequantised values with a one-step tolerance and check that the scales are the oracle's for all but near-tie blocks
    o8 = torch.zeros((M, N), dtype=torch.uint8, device=DEV)
    o8s = torch.zeros((M, N // 32), dtype=torch.uint8, device=DEV)
    E.check(lib.f5_op_gemm_f8(P(a8), P(asc), P(w8), P(wsc), P(bias_d), P(None), P(None), P(None), P(None), P(o8), P(o8s),
                              M, N, K, K, K, N, 2, stream()), "gemm_f8 gelu")
    sync()
    g = F.gelu(y, approximate="tanh")
    q_ref, s_ref = MX.mx_quantize(g.float())
    same = (o8s.cpu() == s_ref).float().mean()
    print(f"   gelu->fp8: {100 * float(same):.2f} % of block scales equal the oracle's")
    assert same > 0.995
    got = MX.mx_dequantize(o8.cpu(), o8s.cpu())
    step = torch.pow(2.0, o8s.cpu().double() - 127 + 5).repeat_interleave(32, dim=1)     # top-binade step of the block
    assert torch.all((got - g).abs() <= 0.51 * step + 1e-6)
