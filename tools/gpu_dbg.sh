#!/bin/bash
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
for b in 8 16 24 32; do
  timeout 300 python bench.py --batch $b --steps 3 --warmup 1 --no-sub --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($b, d['value'], d['ms_per_step'], d.get('roofline',{}).get('frac'))" | tee -a $OUT/batch_sweep.txt
done
