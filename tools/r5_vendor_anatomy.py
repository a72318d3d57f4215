"""Round-5 measurement (nothing of it enters the product): what IS the vendor GEMM that `tools/yardstick.py` times next to ours?

  run        (under `rocprofv3 --kernel-trace`, optionally `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES`)
             torch.matmul (hipBLASLt / rocBLAS) and our f5_op_* launch on the four DiT-block shapes at M = 59 968, fp16,
             workload-like operands, 6 launches each; then torch SDPA and f5_op_attention
  summarize  DIR  reads rocprofv3's *_kernel_trace.csv (and *_counter_collection.csv when present) under DIR and prints, per kernel
             name: launches, grid / workgroup size, LDS bytes, VGPR / AGPR / SGPR, scratch, average duration, TF/s for the GEMM shapes,
             waves per CU implied by the resources, MFMA-busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 1024 SIMDs)

usage (GPU box): cd /tmp && rocprofv3 --kernel-trace --output-format csv -d OUT -o va -- python $R/tools/r5_vendor_anatomy.py run
                 python tools/r5_vendor_anatomy.py summarize OUT > profiles/r05/vendor_gemm_anatomy.txt
"""
import csv
import glob
import json
import os
import sys

M_ROWS, D, FF, H, N_FRAMES = 59968, 1024, 2048, 16, 937
SHAPES = [("qkv", 3 * D, D), ("out_proj", D, D), ("ff1", FF, D), ("ff2", D, FF)]


def run():
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from f5_tts_mlx_amd import engine as E
    lib = E.load_library()
    dev = torch.device("cuda:0")
    P = E.ptr
    st = lambda: E.stream_ptr(dev)   # noqa: E731
    opd = torch.float16
    g = torch.Generator(device="cpu").manual_seed(0)
    mk = lambda std, *s: (torch.randn(*s, generator=g) * std).to(dev).to(opd)   # noqa: E731
    with E.operand_type("f16"):
        for name, N, K in SHAPES:
            a, w = mk(1.0, M_ROWS, K), mk(K ** -0.5, N, K)
            out16 = torch.empty(M_ROWS, N, dtype=opd, device=dev)
            wt = w.t()
            for _ in range(6):
                torch.matmul(a, wt, out=out16)
            torch.cuda.synchronize()
            bias = torch.zeros(N, device=dev)
            if name in ("out_proj", "ff2"):
                gate, xres = torch.full((N,), 0.5, device=dev), torch.zeros(M_ROWS, D, device=dev)
                for _ in range(6):
                    E.check(lib.f5_op_gemm_resid_gate(P(a), P(None), P(w), P(None), P(bias), P(gate), P(None), P(xres), M_ROWS, D, K, K, K, D, 1, st()))
            elif name == "ff1":
                for _ in range(6):
                    E.check(lib.f5_op_gemm(P(a), P(None), P(w), P(None), P(bias), P(None), P(out16), P(None), M_ROWS, N, K, K, K, N, 1, 2, st()))
            torch.cuda.synchronize()
            del a, w, out16
        B = 64
        npad = (N_FRAMES + 63) // 64 * 64
        qk = (torch.randn(B * N_FRAMES, 2 * D, generator=g) * 0.6).to(dev).to(opd)
        vt = torch.randn(B * H, 64, npad, generator=g).to(dev).to(opd)
        vt[:, :, N_FRAMES:] = 0
        q4 = qk[:, :D].reshape(B, N_FRAMES, H, 64).transpose(1, 2).contiguous()
        k4 = qk[:, D:].reshape(B, N_FRAMES, H, 64).transpose(1, 2).contiguous()
        v4 = vt[:, :, :N_FRAMES].reshape(B, H, 64, N_FRAMES).transpose(2, 3).contiguous()
        for _ in range(4):
            torch.nn.functional.scaled_dot_product_attention(q4, k4, v4, scale=1.0)
        torch.cuda.synchronize()
    print("done")


def summarize(d):
    rows = []
    for f in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(f) as fh:
            rows += list(csv.DictReader(fh))
    ctr = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name", "")
                ctr.setdefault(k, {}).setdefault(r.get("Counter_Name", ""), []).append(float(r.get("Counter_Value", 0) or 0))
    by = {}
    for r in rows:
        name = r.get("Kernel_Name", "")
        e = by.setdefault(name, dict(n=0, dur=[], r=r))
        e["n"] += 1
        try:
            e["dur"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        except (KeyError, ValueError):
            pass
    out = []
    for name, e in sorted(by.items(), key=lambda kv: -sum(kv[1]["dur"])):
        r = e["r"]
        gx = lambda k: r.get(k, "?")   # noqa: E731
        dur = sorted(e["dur"])
        rec = dict(kernel=name[:260], launches=e["n"], avg_us=round(sum(dur) / max(1, len(dur)), 1), min_us=round(dur[0], 1) if dur else None,
                   grid=[gx("Grid_Size_X"), gx("Grid_Size_Y"), gx("Grid_Size_Z")], workgroup=[gx("Workgroup_Size_X"), gx("Workgroup_Size_Y"), gx("Workgroup_Size_Z")],
                   lds_bytes=gx("LDS_Block_Size"), scratch=gx("Scratch_Size"), vgpr=gx("VGPR_Count"), agpr=gx("Accum_VGPR_Count"), sgpr=gx("SGPR_Count"))
        try:
            wg = int(rec["workgroup"][0]) * int(rec["workgroup"][1]) * int(rec["workgroup"][2])
            regs = int(rec["vgpr"]) + int(rec["agpr"])
            alloc = (regs + 7) // 8 * 8
            waves_simd = min(8, 512 // max(alloc, 1))
            lds = int(rec["lds_bytes"])
            wg_by_lds = 163840 // lds if lds > 0 else 99
            wg_by_regs = waves_simd * 4 * 64 // wg
            rec["workgroups_per_cu"] = min(wg_by_lds, wg_by_regs, 2048 // wg)
            rec["n_workgroups"] = int(rec["grid"][0]) * int(rec["grid"][1]) * int(rec["grid"][2]) // wg
        except (ValueError, ZeroDivisionError):
            pass
        c = ctr.get(name)
        if c and "SQ_VALU_MFMA_BUSY_CYCLES" in c and "GRBM_GUI_ACTIVE" in c:
            rec["mfma_busy"] = round(sum(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (sum(c["GRBM_GUI_ACTIVE"]) * 1024.0), 3)
        out.append(rec)
    for rec in out:
        print(json.dumps(rec))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "summarize":
        summarize(sys.argv[2])
    else:
        run()
