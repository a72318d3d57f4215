"""Round-5 numerics study of the LN fold (CPU, fp64 reference; no GPU needed): how far is

    (LN(x)(1 + s) + b) W^T      with LayerNorm folded behind the GEMM (engine option "ln_fold", csrc/gemm.hpp fold_*)

from the exact value, against the same through the UNFOLDED path (LN in fp32, its output rounded to the 16-bit operand), as a function
of the row mean |mu| / sigma, with and without massive-activation channels, for

    unfolded                 h16 = round(LN(x)(1 + s) + b),  out = h16 W16^T
    round 4 (unshifted)      A = round(x (1 + s)),  out = rstd (A W16^T) - rstd mu c1 + c2;  variance from one-pass fp32 sums E[x^2] - E[x]^2
    unshifted, exact stats   the same with mean / variance computed exactly (isolates the operand rounding from the variance cancellation)
    round 5 (shifted)        A = round((x - m)(1 + s)) with m = the row's mean at the previous LayerNorm = mu - drift;
                             out = rstd (A W16^T) - rstd (mu - m) c1 + c2;  slice statistics merged with Chan's update (fp32)

All GEMMs are evaluated in fp64 on the rounded operands (the MFMA accumulates in fp32: its own error is common to all variants).
Error measure: mean |out - exact| / rms(exact).  Prints one JSON line per configuration and a break-point summary:
the |mu| / sigma at which a variant's error exceeds 2x the unfolded path's.

usage: python tools/r5_fold_numerics_study.py [--op f16|bf16] > profiles/r05/ln_fold_numerics_study.jsonl
"""
import argparse
import json

import numpy as np
import torch


def rnd(x, op):
    return x.to(torch.float16 if op == "f16" else torch.bfloat16).double()


def chan_stats(d32, eps):
    """fp32 restatement of the kernels: per 64-column slice (sum, centred M2), merged slice by slice (f5_fold_rows_kernel)"""
    M, D = d32.shape
    sl = d32.reshape(M, D // 64, 64)
    s = sl.sum(-1, dtype=torch.float32)
    ms = s / 64.0
    m2s = ((sl - ms[..., None]) ** 2).sum(-1, dtype=torch.float32)
    mean = torch.zeros(M, dtype=torch.float32)
    m2 = torch.zeros(M, dtype=torch.float32)
    for i in range(D // 64):
        delta = ms[:, i] - mean
        inv = 1.0 / (i + 1)
        mean = mean + delta * inv
        m2 = m2 + m2s[:, i] + delta * delta * (64.0 * i * inv)
    rstd = torch.rsqrt(m2 / D + eps)
    return rstd.double()[:, None], mean.double()[:, None]


def study(op, mu_over_sigma, outliers, drift, M=512, D=1024, NO=256, seed=0, eps=1e-6):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, D, generator=g, dtype=torch.float64)
    if outliers:
        x[:, :4] *= 1.0e3                                              # four massive-activation channels
    x = x - x.mean(-1, keepdim=True)
    sigma = x.std(-1, unbiased=False, keepdim=True)
    sign = torch.where(torch.rand(M, 1, generator=g, dtype=torch.float64) < 0.5, -1.0, 1.0)
    mu = mu_over_sigma * sigma * sign
    x = (x + mu).float()                                               # the fp32 residual stream
    s = 0.3 * torch.randn(D, generator=g, dtype=torch.float64)
    b = 0.3 * torch.randn(D, generator=g, dtype=torch.float64)
    W = torch.randn(NO, D, generator=g, dtype=torch.float64) * D ** -0.5
    W16 = rnd(W, op)
    x64 = x.double()
    mean = x64.mean(-1, keepdim=True)
    var = ((x64 - mean) ** 2).mean(-1, keepdim=True)
    rstd = torch.rsqrt(var + eps)
    h = (x64 - mean) * rstd * (1 + s) + b
    exact = h @ W16.T
    rms = float(exact.pow(2).mean().sqrt())
    err = lambda o: float((o - exact).abs().mean()) / rms   # noqa: E731
    # unfolded: LN in fp32 (two-pass, as ln_modulate_kernel), output rounded
    m32 = x.mean(-1, keepdim=True)
    v32 = ((x - m32) ** 2).mean(-1, keepdim=True)
    h32 = (x - m32) * torch.rsqrt(v32 + eps) * (1 + s.float()) + b.float()
    out = {"unfolded": err(rnd(h32, op) @ W16.T)}
    c1 = (1 + s) @ W16.T
    c2 = b @ W16.T
    # round 4: unshifted operand, one-pass fp32 sums
    A = rnd(x * (1 + s.float()), op)
    s1 = x.sum(-1, keepdim=True, dtype=torch.float32)
    s2 = (x * x).sum(-1, keepdim=True, dtype=torch.float32)
    mean1 = s1 / D
    var1 = torch.clamp(s2 / D - mean1 * mean1, min=0.0)
    rstd1 = torch.rsqrt(var1 + eps).double()
    out["r4_unshifted_onepass"] = err(rstd1 * (A @ W16.T) - rstd1 * mean1.double() * c1 + c2)
    out["unshifted_exact_stats"] = err(rstd * (A @ W16.T) - rstd * mean * c1 + c2)
    # round 5: shifted operand (m = previous mean = mu - drift * sigma), Chan-merged slice statistics in fp32
    m = (mean - drift * sigma).float()
    d32 = x - m
    A5 = rnd(d32 * (1 + s.float()), op)
    rstd5, mean_d = chan_stats(d32, eps)
    out["r5_shifted"] = err(rstd5 * (A5 @ W16.T) - rstd5 * mean_d * c1 + c2)
    out["operand_peak_unshifted"] = float((x * (1 + s.float())).abs().max())
    out["operand_peak_shifted"] = float((d32 * (1 + s.float())).abs().max())
    return out


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--op", default="f16", choices=["f16", "bf16"])
    ns = ap.parse_args()
    grid = [0.0, 0.5, 1.0, 3.0, 10.0, 30.0, 100.0, 1000.0]
    summary = {}
    for outliers in (False, True):
        for drift in (0.0, 0.3, 1.0, 3.0):
            rows = []
            for r_ in grid:
                o = study(ns.op, r_, outliers, drift)
                rows.append(o)
                print(json.dumps(dict(op=ns.op, mu_over_sigma=r_, outlier_channels=outliers, drift_sigma=drift, **{k: (round(v, 7) if v < 1 else round(v, 1)) for k, v in o.items()})), flush=True)
            for k in ("r4_unshifted_onepass", "unshifted_exact_stats", "r5_shifted"):
                bp = next((g_ for g_, o in zip(grid, rows) if o[k] > 2.0 * o["unfolded"]), None)
                summary[f"{k} outliers={outliers} drift={drift}"] = bp
    print(json.dumps(dict(op=ns.op, breakpoint_mu_over_sigma_where_error_exceeds_2x_unfolded=summary)))
