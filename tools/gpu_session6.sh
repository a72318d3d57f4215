#!/bin/bash
TAG=${1:-s6}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; tail -4 $OUT/pytest.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_b1.json 2> $OUT/bench_b1.err; cut -c1-200 $OUT/bench_b1.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_b32 -o b32 -- python $R/bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline > $R/$OUT/prof_b32.json 2> $R/$OUT/prof_b32.err
cd $R
cut -c1-200 $OUT/prof_b32.json
find $OUT/prof_b32 -name "*kernel_trace.csv" -delete
python - <<'PY'
import csv,glob
f=glob.glob("gpurun_out/'$TAG'/prof_b32/*kernel_stats.csv".replace("'",""))[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:9]:
    print(f"{r['Name'][:56]:56s} calls={r['Calls']:>6s} {100*float(r['TotalDurationNs'])/tot:5.1f}% avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
