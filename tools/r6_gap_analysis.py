"""Sum of kernel durations vs the gaps between consecutive kernels of a rocprofv3 --kernel-trace CSV (one stream): where does the wall time of
a sample() go that the per-kernel averages do not show?  usage: python tools/r6_gap_analysis.py <kernel_trace.csv> [skip_first_n]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows), key=lambda t: t[0])[skip:]
ev = [e for e in ev if "mfma_peak" not in e[2]]
dur = sum(e[1] - e[0] for e in ev)
gaps = [max(0, ev[i + 1][0] - ev[i][1]) for i in range(len(ev) - 1)]
small = [g for g in gaps if g < 100_000]          # (gaps above 100 us are boundaries between calls / host work)
span = ev[-1][1] - ev[0][0]
by = {}
for i, g in enumerate(gaps):
    if g < 100_000:
        k = ev[i][2][:60] + " -> " + ev[i + 1][2][:60]
        a = by.setdefault(k, [0, 0])
        a[0] += 1
        a[1] += g
print(f"kernels {len(ev)}  span {span / 1e6:.2f} ms  sum of durations {dur / 1e6:.2f} ms  gaps < 100 us: {sum(small) / 1e6:.2f} ms over {len(small)} ({sum(small) / max(1, len(small)) / 1e3:.2f} us each)")
for k, (n, t) in sorted(by.items(), key=lambda kv: -kv[1][1])[:12]:
    print(f"   {t / n / 1e3:6.2f} us x {n:5d}  {k}")
