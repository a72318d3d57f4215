#!/bin/bash
# usage: tools/gpu_prof.sh TAG "<bench args>"  -> kernel stats csv under gpurun_out/TAG/
TAG=$1; shift
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof -o p -- python $R/bench.py "$@" --no-cpu-baseline > $R/$OUT/bench.json 2> $R/$OUT/bench.err
cd $R
find $OUT/prof -name "*kernel_trace.csv" -delete
cut -c1-300 $OUT/bench.json
python - <<PY
import csv,glob
f=glob.glob('$OUT/prof/*kernel_stats.csv')[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:26]:
    print(f"{r['Name'][:60]:60s} calls={r['Calls']:>6s} {100*float(r['TotalDurationNs'])/tot:5.1f}% avg_us={float(r['AverageNs'])/1e3:9.1f}")
PY
