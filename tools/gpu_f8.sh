#!/bin/bash
# same-box comparison: bf16 vs mxfp8 at B=1 / B=32 (32-point Euler) and BASELINE configs[4] (c5)
OUT=gpurun_out/${1:-f8}
mkdir -p $OUT
run() { python bench.py "$@" --no-cpu-baseline 2>$OUT/err.txt | tail -1; }
run --steps 5 --warmup 2 > $OUT/b1_bf16.json
run --steps 5 --warmup 2 --precision mxfp8 > $OUT/b1_mxfp8.json
run --batch 32 --steps 2 --warmup 1 > $OUT/b32_bf16.json
run --batch 32 --steps 2 --warmup 1 --precision mxfp8 > $OUT/b32_mxfp8.json
run --config c5 --steps 2 --warmup 1 > $OUT/c5.json
run --batch 32 --steps 2 --warmup 1 --method midpoint --ode-points 16 --vocoder > $OUT/c5_bf16.json
for f in b1_bf16 b1_mxfp8 b32_bf16 b32_mxfp8 c5 c5_bf16; do python - <<PY
import json
try:
    d=json.load(open('$OUT/$f.json'))
    print('$f', 'ms', round(d['ms_per_step'],1), 'frames/s', round(d['value']), 'rtf', round(d['rtf'],1), 'roof', round(d['roofline']['achieved']), d['roofline']['unit'], round(d['roofline']['frac'],3))
except Exception as e:
    print('$f', 'FAILED', e); print(open('$OUT/err.txt').read()[-800:])
PY
done
