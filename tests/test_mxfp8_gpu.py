"""MX-fp8 path (BASELINE configs[4]): quantiser, 256x256x128 MX GEMM and its epilogues against the MX oracle."""
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


def test_quantize_mx_golden(lib):
    """the device quantiser against the committed golden bytes / scales (tests/golden/mx_quantize.npz)"""
    import os
    m = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "mx_quantize.npz"))
    q, sc = _dev_quantize(lib, torch.from_numpy(m["x"]))
    assert np.array_equal(q.cpu().numpy(), m["q"]) and np.array_equal(sc.cpu().numpy(), m["e8"])


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


# ------------------------------------------------------------------------------------------------
# model level: engine precision "mxfp8" vs the oracle with the same MX operand rounding
# ------------------------------------------------------------------------------------------------
from f5test import O, TINY, F5TTS_335M, synth_inputs, synthetic_weights   # noqa: E402
from f5_tts_mlx_amd.cfm import F5TTS                                      # noqa: E402
from f5_tts_mlx_amd.dit import DiT                                        # noqa: E402


def _pad_cond(cond, N):
    B, n_ref, mel = cond.shape
    out = torch.zeros((B, N, mel))
    out[:, :n_ref] = cond
    return out


@pytest.mark.parametrize("B,N,ragged", [(1, 150, False), (2, 200, True)])
def test_dit_forward_mxfp8(B, N, ragged):
    cfg = TINY
    w = synthetic_weights(cfg, seed=42)
    cond, text, durations, y0 = synth_inputs(cfg, B, N, nt=24, n_ref=30, seed=B * 100 + N, ragged=ragged)
    sc = _pad_cond(cond, N)
    mask = O.lens_to_mask(torch.tensor(durations), N) if B > 1 else None
    t = 0.37
    m = DiT.from_config(cfg, precision="mxfp8", device=DEV)
    m.load_weights(w)
    pred, null = m.engine.dit_forward(y0.to(DEV), text.to(DEV), sc.to(DEV), [N] * B, durations, t)
    torch.cuda.synchronize()
    emu = O.DiTOracle(cfg, w, emulate_mxfp8=True)
    fp32 = O.DiTOracle(cfg, w)
    for name, got, flags in (("cond", pred, (False, False)), ("null", null, (True, True))):
        want = emu.forward(y0, sc, text, torch.tensor(t), flags[0], flags[1], mask)
        full = fp32.forward(y0, sc, text, torch.tensor(t), flags[0], flags[1], mask)
        _, mean, refm = report(f"mxfp8 forward [{name}] B{B} N{N} vs oracle[mxfp8 emulation]", got.cpu(), want)
        _, drift, _ = report(f"mxfp8 forward [{name}] drift vs oracle[fp32]", got.cpu(), full)
        # same operand rounding on both sides: what remains are fp8 rounding flips caused by fp32-level differences upstream
        assert mean <= 1.5e-2 * max(1.0, refm)
        assert drift <= 0.15 * max(1.0, refm)


def test_sample_mxfp8_midpoint_with_vocoder():
    """BASELINE configs[4] in miniature: MX-fp8 weights/activations, midpoint solver, Vocos on top"""
    from f5_tts_mlx_amd.vocos import Vocos, synthetic_vocos_weights
    cfg = TINY
    w = synthetic_weights(cfg, seed=42)
    B, N = 2, 160
    cond, text, durations, y0 = synth_inputs(cfg, B, N, nt=24, n_ref=40, seed=9)
    m = DiT.from_config(cfg, precision="mxfp8", device=DEV)
    m.load_weights(w)
    voc = Vocos(synthetic_vocos_weights(seed=7), device=DEV)
    tts = F5TTS(m, vocoder=voc.decode)
    mel_model = F5TTS(m)
    out_mel, traj = mel_model.sample(cond.to(DEV), text=text.to(DEV), duration=torch.tensor(durations), steps=5, method="midpoint",
                                     y0=y0.to(DEV))
    torch.cuda.synchronize()
    emu = O.DiTOracle(cfg, w, emulate_mxfp8=True)
    want, _ = O.sample(emu, cond, text, torch.tensor(durations), steps=5, method="midpoint", y0=y0)
    _, mean, refm = report("mxfp8 5-point midpoint sample vs oracle[mxfp8 emulation]", out_mel.cpu(), want)
    assert mean <= 3e-2 * max(1.0, refm)
    wave, _ = tts.sample(cond.to(DEV), text=text.to(DEV), duration=torch.tensor(durations), steps=5, method="midpoint", y0=y0.to(DEV))
    torch.cuda.synchronize()
    assert wave.shape[0] == B and torch.isfinite(wave).all()
    # graph replay == eager, bitwise
    a, _ = mel_model.sample(cond.to(DEV), text=text.to(DEV), duration=torch.tensor(durations), steps=5, method="midpoint", y0=y0.to(DEV),
                            use_graph=False)
    torch.cuda.synchronize()
    assert torch.equal(a, out_mel)


def test_full_size_forward_mxfp8_drift():
    """335M, N=937: one CFG evaluation; reports the drift of the MX-fp8 mode against the fp32 oracle"""
    cfg = F5TTS_335M
    w = synthetic_weights(cfg, seed=42)
    N = 937
    cond, text, durations, y0 = synth_inputs(cfg, 1, N, nt=160, n_ref=281, seed=1)
    sc = _pad_cond(cond, N)
    m = DiT.from_config(cfg, precision="mxfp8", device=DEV)
    m.load_weights(w)
    pred, null = m.engine.dit_forward(y0.to(DEV), text.to(DEV), sc.to(DEV), [N], durations, 0.25)
    torch.cuda.synchronize()
    emu = O.DiTOracle(cfg, w, emulate_mxfp8=True)
    want = emu.forward(y0, sc, text, torch.tensor(0.25), False, False, None)
    full = O.DiTOracle(cfg, w).forward(y0, sc, text, torch.tensor(0.25), False, False, None)
    _, mean, refm = report("full-size mxfp8 forward vs oracle[mxfp8 emulation]", pred.cpu(), want)
    _, drift, _ = report("full-size mxfp8 forward drift vs oracle[fp32]", pred.cpu(), full)
    assert mean <= 3e-2 * max(1.0, refm)
    assert drift <= 0.3 * max(1.0, refm)
