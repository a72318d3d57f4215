"""Round-5 probe (GPU box, measurement build only): the timeline of every workgroup of ONE 256x256 GEMM launch, from wall-clock
stamps the kernel leaves behind (csrc/gemm256.hip F5_PROBE_TS: start, prologue staged, K loop done, epilogue done per wave group,
HW_ID, XCC_ID; 100 MHz clock = 10 ns).  Per variant (full / main loop only / epilogue without stores / without arithmetic):

  * K-loop duration per workgroup (median, p10, p90, by round): does the main loop itself slow down when the epilogue runs?
  * epilogue duration per workgroup and wave group
  * per CU: the gap between one workgroup's end and the next one's start, and between its end and the next one's first K step
  * how far apart the CUs of one XCD are in time (de-phasing), round by round

usage: F5TTS_HIP_LIB=f5_tts_mlx_amd/csrc/libf5tts_hip_probe.so python tools/r5_timeline_probe.py [shape ...] > timeline.jsonl
"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from f5_tts_mlx_amd import engine as E  # noqa: E402

lib = E.load_library()
dev = torch.device("cuda:0")
P = E.ptr
st = lambda: E.stream_ptr(dev)   # noqa: E731
M_ROWS, D, FF, H, N_FRAMES = 59968, 1024, 2048, 16, 937
MAXWG = 4096
FOLD = False


def pct(a, q):
    return round(float(np.percentile(a, q)), 2)


def timeline(nwg):
    buf = (C.c_ulonglong * (MAXWG * 8))()
    assert lib.f5_probe_read_ts(buf, MAXWG * 8, 1) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(MAXWG, 8)[:nwg].astype(np.int64)
    return t


def analyse(t, has_epi, vsplit=None):
    us = lambda x: x / 100.0   # noqa: E731   10 ns ticks -> us
    t0 = t[:, 0].min()
    start, staged, kdone0, kdone4 = us(t[:, 0] - t0), us(t[:, 1] - t0), us(t[:, 2] - t0), us(t[:, 7] - t0)
    rec = {}
    kdur = kdone0 - start
    rec["k_loop_us"] = dict(median=pct(kdur, 50), p10=pct(kdur, 10), p90=pct(kdur, 90))
    rec["prologue_us"] = dict(median=pct(staged - start, 50), p90=pct(staged - start, 90))
    rec["k_loop_steady_us"] = dict(median=pct(kdone0 - staged, 50), p10=pct(kdone0 - staged, 10), p90=pct(kdone0 - staged, 90))
    end = np.maximum(kdone0, kdone4)
    if has_epi:
        e0, e4 = us(t[:, 3] - t0), us(t[:, 4] - t0)
        ok = (t[:, 3] > 0) & (t[:, 4] > 0)
        rec["epilogue_wave0_us"] = dict(median=pct((e0 - kdone0)[ok], 50), p10=pct((e0 - kdone0)[ok], 10), p90=pct((e0 - kdone0)[ok], 90))
        rec["epilogue_wave4_us"] = dict(median=pct((e4 - kdone4)[ok], 50), p10=pct((e4 - kdone4)[ok], 10), p90=pct((e4 - kdone4)[ok], 90))
        end = np.where(ok, np.maximum(e0, e4), end)
    rec["launch_span_us"] = round(float(end.max()), 1)
    # per CU: (xcc, se, sh, cu) from HW_ID; successive workgroups ordered by start time
    hw, xcc, tn = t[:, 5], t[:, 6] & 0xF, (t[:, 6] >> 8) & 0xFFF
    cu_key = (xcc << 16) | (hw & 0xFF00)              # cu_id[11:8], sh_id[12], se_id[15:13]
    gaps, first_k_gaps, per_cu = [], [], {}
    for k in np.unique(cu_key):
        idx = np.where(cu_key == k)[0]
        idx = idx[np.argsort(start[idx])]
        per_cu[int(k)] = len(idx)
        for a, b in zip(idx[:-1], idx[1:]):
            gaps.append(start[b] - end[a])
            first_k_gaps.append(staged[b] - end[a])
    rec["cus_seen"] = len(per_cu)
    rec["wg_per_cu"] = dict(min=min(per_cu.values()), max=max(per_cu.values()))
    if gaps:
        rec["gap_end_to_next_start_us"] = dict(median=pct(gaps, 50), p90=pct(gaps, 90))
        rec["gap_end_to_next_staged_us"] = dict(median=pct(first_k_gaps, 50), p90=pct(first_k_gaps, 90))
    # de-phasing: spread of the start times of the r-th workgroup of every CU, per XCD
    spread = {}
    for r in range(0, 12):
        s_r = []
        for x in range(8):
            st_r = []
            for k in np.unique(cu_key[xcc == x]):
                idx = np.where(cu_key == k)[0]
                idx = idx[np.argsort(start[idx])]
                if len(idx) > r:
                    st_r.append(start[idx[r]])
            if len(st_r) > 4:
                s_r.append(float(np.percentile(st_r, 90) - np.percentile(st_r, 10)))
        if s_r:
            spread[r] = round(float(np.mean(s_r)), 2)
    rec["start_spread_p10_p90_within_xcd_by_round_us"] = spread
    # K-loop duration by round
    byr = {}
    order = np.argsort(start)
    for r in range(0, 12):
        sel = order[r * 256:(r + 1) * 256]
        if len(sel) > 32:
            byr[r] = pct(kdur[sel], 50)
    rec["k_loop_median_by_round_us"] = byr
    # by the column tile of the workgroup itself and of its predecessor on the same CU (QKV: tn >= 8 are the transposed V tiles)
    if vsplit is not None:
        pred_v = np.zeros(len(start), dtype=np.int64) - 1
        for k in np.unique(cu_key):
            idx = np.where(cu_key == k)[0]
            idx = idx[np.argsort(start[idx])]
            for a, b in zip(idx[:-1], idx[1:]):
                pred_v[b] = 1 if tn[a] >= vsplit else 0
        isv = tn >= vsplit
        rec["k_loop_median_us_own_tile_qk_vs_v"] = [pct(kdur[~isv], 50), pct(kdur[isv], 50)]
        rec["k_loop_median_us_after_qk_vs_after_v"] = [pct(kdur[pred_v == 0], 50), pct(kdur[pred_v == 1], 50)] if (pred_v == 1).any() and (pred_v == 0).any() else None
        if has_epi:
            ep = np.maximum(us(t[:, 3] - t0) - kdone0, us(t[:, 4] - t0) - kdone4)
            okm = (t[:, 3] > 0) & (t[:, 4] > 0)
            rec["epilogue_median_us_qk_vs_v"] = [pct(ep[okm & ~isv], 50), pct(ep[okm & isv], 50)]
    return rec


def main():
    assert "probe" in str(E.library_path()), "load the measurement build through F5TTS_HIP_LIB"
    lib.f5_probe_read_ts.argtypes = [C.c_void_p, C.c_int, C.c_int]
    opd = torch.float16
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda std, *s: (torch.randn(*s, generator=g) * std).to(dev).to(opd)   # noqa: E731
    shapes = [("qkv", 3 * D, D, "qkv"), ("out_proj", D, D, "resid"), ("ff1", FF, D, "gelu"), ("ff2", D, FF, "resid")]
    global FOLD
    only = [a for a in sys.argv[1:] if a != "--fold"]
    FOLD = "--fold" in sys.argv
    with E.operand_type("f16"):
        for name, N, K, kind in shapes:
            if only and name not in only:
                continue
            a, w = mk(1.0, M_ROWS, K), mk(K ** -0.5, N, K)
            bias = torch.zeros(N, device=dev)
            gate = torch.full((N,), 0.5, device=dev)
            xres = torch.zeros(M_ROWS, D, device=dev)
            out16 = torch.empty(M_ROWS, N, dtype=opd, device=dev)
            npad = (N_FRAMES + 63) // 64 * 64
            cos_t, sin_t = torch.ones(N_FRAMES, 32, device=dev), torch.zeros(N_FRAMES, 32, device=dev)
            tt = [torch.empty(64 * N_FRAMES, device=dev) for _ in range(2)]
            E.check(lib.f5_op_rope_table(P(cos_t), P(sin_t), N_FRAMES, 64, st()))
            E.check(lib.f5_op_rope_table_g4(P(tt[0]), P(tt[1]), N_FRAMES, 64, C.c_float(1.0), st()))
            qk = torch.empty(M_ROWS, 2 * D, dtype=opd, device=dev)
            vt = torch.zeros(64 * H, 64, npad, dtype=opd, device=dev)
            if kind == "resid":
                ours = lambda: E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bias), P(gate), P(None), P(xres),   # noqa: E731
                                                                 M_ROWS, D, K, K, K, D, 1, st()))
            elif kind == "gelu":
                ours = lambda: E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(out16), P(None), M_ROWS, N, K, K, K, N, 1, 2, st()))   # noqa: E731
            else:
                E.check(lib.f5_debug_set_op_rope_tables_g4(P(tt[0]), P(tt[1])))
                ours = lambda: E.check(lib.f5_op_qkv_rope(P(a), P(None), P(w), P(None), P(bias), P(cos_t), P(sin_t), P(qk), P(None), P(vt),   # noqa: E731
                                                          P(None), 64, N_FRAMES, npad, H, D, 1, st()))
            nwg = min(MAXWG, ((M_ROWS + 255) // 256) * (N // 256))
            keep = []
            if FOLD:          # the launches sample() makes at this size: LN fold consumer (QKV / FF1) or producer (residual GEMMs)
                if kind == "resid":
                    x16, stats = torch.empty(M_ROWS, D, dtype=opd, device=dev), torch.zeros(D // 64, M_ROWS, 2, device=dev)
                    sc, sh = torch.zeros(D, device=dev), torch.zeros(M_ROWS, device=dev)
                    keep += [x16, stats, sc, sh]
                    E.check(lib.f5_debug_set_op_fold_producer(P(sc), P(x16), P(stats), P(sh)))
                else:
                    rowf = torch.ones(M_ROWS, 2, device=dev)
                    c1, c2 = torch.zeros(N, device=dev), torch.zeros(N, device=dev)
                    keep += [rowf, c1, c2]
                    E.check(lib.f5_debug_set_op_fold_consumer(P(rowf), P(c1), P(c2)))
            variants = [("full", 0), ("main_loop", 1), ("no_store", 0x18000), ("no_math", 0x20000), ("no_store_no_math", 0x38000)]
            if kind == "qkv":
                variants += [("no_qk_store", 0x10000), ("no_v_store", 0x8000)]
            for key, flags in variants:
                lib.f5_debug_set_gemm_flags(flags)
                for _ in range(3):
                    ours()
                torch.cuda.synchronize()
                # 20 launches back to back; the stamps left behind are those of the LAST one (every launch overwrites the same slots):
                # a launch timed on its own after a host synchronisation runs 20-25 % slower (clock ramp), this one is in steady state
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(20):
                    ours()
                e1.record()
                torch.cuda.synchronize()
                t = timeline(nwg)
                rec = dict(kind="timeline", shape=name, fold=FOLD, variant=key, launch_us=round(e0.elapsed_time(e1) * 1e3 / 20, 1), workgroups=nwg)
                rec.update(analyse(t, has_epi=(flags & 1) == 0, vsplit=8 if kind == "qkv" else None))
                print(json.dumps(rec), flush=True)
            lib.f5_debug_set_gemm_flags(0)
            E.check(lib.f5_debug_set_op_fold_producer(P(None), P(None), P(None), P(None)))
            E.check(lib.f5_debug_set_op_fold_consumer(P(None), P(None), P(None)))
            if kind == "qkv":
                E.check(lib.f5_debug_set_op_rope_tables_g4(P(None), P(None)))


if __name__ == "__main__":
    main()
