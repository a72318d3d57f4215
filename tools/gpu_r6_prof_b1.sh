#!/bin/bash
# rocprofv3 kernel stats of the batch-1 bench run (hipGraph) -- per-kernel averages with the batch-1-sized LN fold on (default) and off
TAG=${1:-r6p}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for F in on off; do
  X=""; [ $F = off ] && X="F5_BENCH_LN_FOLD=0"
  env $X timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_$F -o b1 -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-sub > $R/$OUT/prof_$F.json 2> $R/$OUT/prof_$F.err
  f=$(find $R/$OUT/prof_$F -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $R/$OUT/b1_fold_${F}_kernel_stats.csv
done
cd $R
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +4M -delete
python - <<P
import csv
for F in ("off", "on"):
    rows = [r for r in csv.DictReader(open("$OUT/b1_fold_%s_kernel_stats.csv" % F)) if "mfma_peak" not in r["Name"] and "rocclr" not in r["Name"]]
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print("fold", F, "kernel ms", round(tot / 1e6, 1))
    for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:8]:
        print("   %-95s calls %6d avg %7.2f us share %5.1f%%" % (r["Name"][:95], int(r["Calls"]), float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / tot * 100))
P
