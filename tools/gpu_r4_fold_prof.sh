#!/bin/bash
# per-kernel durations of sample() at batch 32 with the LN fold off / on (rocprofv3 kernel trace)
TAG=${1:-r4foldprof}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for opt in 0 1; do
  R=$(pwd)
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof$opt -o p -- python $R/tools/r4_ln_fold_ab.py --batches ${B:-32} --only $opt --steps 4 > $R/$OUT/prof$opt.log 2>&1)
  f=$(find /tmp/prof$opt -name "*kernel_stats.csv" | head -1)
  tail -3 $OUT/prof$opt.log
  cp "$f" $OUT/kernel_stats_fold$opt.csv
  echo "== ln_fold=$opt"; head -12 $OUT/kernel_stats_fold$opt.csv | cut -c1-200
done
