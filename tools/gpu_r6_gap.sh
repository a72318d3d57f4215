#!/bin/bash
TAG=${1:-r6gap}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for F in 0 -1; do
  F5_BENCH_LN_FOLD=$F timeout 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/tr_$F -o b1 -- python $R/bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-sub > $R/$OUT/tr_$F.json 2> $R/$OUT/tr_$F.err
  f=$(find $R/$OUT/tr_$F -name "*kernel_trace.csv" | head -1)
  echo "== ln_fold $F"; python $R/tools/r6_gap_analysis.py $f | tee $R/$OUT/gaps_fold_$F.txt
  python -c "import json;r=json.loads(open('$R/$OUT/tr_$F.json').read().strip().split('\n')[-1]);print('ms_per_step', r['ms_per_step'])"
done
find $R/$OUT -name "*kernel_trace.csv" -delete
