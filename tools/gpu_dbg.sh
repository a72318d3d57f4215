#!/bin/bash
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
for rep in 1 2; do
for v in 0 4; do
  for b in 1 32; do
  timeout 300 python tools/bench_flags.py attnvar=$v -- --batch $b --steps 4 --warmup 2 --no-sub --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('attnvar',$v,'batch',$b, d['value'], d['ms_per_step'])" | tee -a $OUT/attnvar_ab.txt
  done
done
done
