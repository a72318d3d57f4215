"""Summarise rocprofv3 --pmc counter_collection CSVs per kernel (mean per dispatch)."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
for d in sorted(glob.glob(os.path.join(out, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    files = glob.glob(os.path.join(d, "*counter_collection.csv"))
    if not files:
        print(d, "no counter csv", os.listdir(d))
        continue
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for row in csv.DictReader(open(files[0])):
        k = row["Kernel_Name"][:48]
        a = agg[k][row["Counter_Name"]]
        a[0] += float(row["Counter_Value"])
        a[1] += 1
    print("==", os.path.basename(d))
    for k, cs in sorted(agg.items(), key=lambda kv: -sum(v[0] for v in kv[1].values()))[:12]:
        print("  %-48s " % k + "  ".join("%s mean=%.4g n=%d" % (c, v[0] / v[1], v[1]) for c, v in sorted(cs.items())))
