"""MFMA utilisation per kernel from one `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE
--kernel-trace` pass over tools/pmc_block_ops.py (the five DiT-block kernels at the bench shape, eager launches).

Per kernel symbol (mean over its launches, the first one of each run dropped):
  mfma_busy_cycles   SQ_VALU_MFMA_BUSY_CYCLES, summed over the chip: 32 cycles per v_mfma_f32_32x32x16 and SIMD
  kernel_cycles      GRBM_GUI_ACTIVE / 8 (one count per XCD)
  mfma_busy_frac     mfma_busy_cycles / (1024 SIMDs x kernel_cycles): the share of SIMD time the matrix pipe is occupied
  duration_us        End - Start of the dispatch in the kernel trace of the SAME pass (profiled passes run a lower clock than
                     un-profiled ones, MI355X_MICROARCH.md "DVFS give-back": compare the fractions, not the times)
  eff_clock_ghz      kernel_cycles / duration: the clock the kernel ran at
usage: python tools/pmc_mfma_summary.py <dir of the pass> <precision> <batch>"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def main():
    d, precision, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    assert cc, "no counter_collection.csv under " + d
    dur = {}
    if kt:
        for r in csv.DictReader(open(kt[0])):
            dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    per = defaultdict(lambda: defaultdict(dict))            # kernel -> dispatch -> counter -> value
    for r in csv.DictReader(open(cc[0])):
        per[r["Kernel_Name"]][int(r["Dispatch_Id"])][r["Counter_Name"]] = float(r["Counter_Value"])
    out = []
    for name, disp in per.items():
        if not any(k in name for k in ("f5_gemm", "f5_attn", "ln_modulate", "mfma_peak", "f5_convpos")):
            continue
        ids = sorted(disp)[1:] or sorted(disp)               # drop the first (cold) launch of a kernel
        mean = lambda c: sum(disp[i].get(c, 0.0) for i in ids) / len(ids)   # noqa: E731
        busy, gui = mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("GRBM_GUI_ACTIVE") / 8.0
        du = [dur[i] for i in ids if i in dur]
        rec = dict(kernel=name[:110], precision=precision, batch=batch, launches=len(ids), mfma_busy_cycles=busy, kernel_cycles=gui,
                   mfma_busy_frac=(busy / (1024.0 * gui)) if gui else None, sq_busy_cycles=mean("SQ_BUSY_CYCLES"),
                   mfma_mops_f16=mean("SQ_INSTS_VALU_MFMA_MOPS_F16"))
        if du:
            rec["duration_us"] = sum(du) / len(du)
            rec["eff_clock_ghz"] = gui / (rec["duration_us"] * 1e3) if rec["duration_us"] else None
        out.append(rec)
    out.sort(key=lambda r: -(r["mfma_busy_cycles"] or 0))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
