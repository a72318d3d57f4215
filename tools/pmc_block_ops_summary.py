"""Turn the counter_collection CSVs of two `rocprofv3 --pmc` passes over tools/pmc_block_ops.py (FETCH_SIZE, WRITE_SIZE) into
entries for profiles/pmc_traffic.json.  Dispatches are grouped in launch order: the runs of f5 GEMM / attention kernels appear
as qkv (1 warm-up + LAUNCHES), attention, out-proj, FF1, FF2; the first launch of each run is dropped (cold L2 / Infinity Cache).
FETCH_SIZE counts 64 B per 128-B request on gfx950 and is doubled (MI355X guide, HBM section); units are KiB.
usage: python tools/pmc_block_ops_summary.py <dir with fetch/ and write/ subdirs> <precision> <batch>"""
import csv
import glob
import json
import os
import sys

KEYS = ["qkv_gemm", "attention", "out_proj_gemm", "ff1_gemm", "ff2_gemm"]


def runs(path, counter):
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    seq = [(r["Kernel_Name"], float(r["Counter_Value"])) for r in rows
           if ("f5_gemm" in r["Kernel_Name"] or "f5_attn" in r["Kernel_Name"])]
    groups = []
    for name, v in seq:
        if groups and groups[-1][0] == name:
            groups[-1][1].append(v)
        else:
            groups.append([name, [v]])
    return groups


def main():
    d, precision, batch = sys.argv[1], sys.argv[2], int(sys.argv[3])
    shapes = {}
    for line in open(os.path.join(d, "driver.txt")):
        p = line.split()
        if p and p[0] in KEYS:
            shapes[p[0]] = (" ".join(p[1:-1]), int(p[-1]))
    f = runs(glob.glob(os.path.join(d, "fetch", "**", "*counter_collection.csv"), recursive=True)[0], "FETCH_SIZE")
    w = runs(glob.glob(os.path.join(d, "write", "**", "*counter_collection.csv"), recursive=True)[0], "WRITE_SIZE")
    assert len(f) == 5 and len(w) == 5, ([g[0] for g in f], [g[0] for g in w])
    out = []
    for key, gf, gw in zip(KEYS, f, w):
        assert gf[0] == gw[0]
        fv, wv = gf[1][1:], gw[1][1:]
        fk, wk = sum(fv) / len(fv), sum(wv) / len(wv)
        out.append(dict(key=key, kernel=gf[0][:100], shape=shapes[key][0], precision=precision, batch=batch, launches_averaged=len(fv),
                        fetch_kib_raw=fk, fetch_bytes_corrected_x2=int(fk * 1024 * 2), write_kib=wk, write_bytes=int(wk * 1024),
                        algorithmic_bytes=shapes[key][1]))
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
