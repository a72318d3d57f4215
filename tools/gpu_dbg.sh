#!/bin/bash
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "band_major" > $OUT/pytest_band.txt 2>&1; grep -E "passed|failed" $OUT/pytest_band.txt | tail -2; grep -E "^(FAILED|ERROR)|assert" $OUT/pytest_band.txt | head
for rep in 1 2; do
for v in 0 1 2; do
  timeout 300 python tools/bench_flags.py order=$v -- --batch 1 --steps 4 --warmup 2 --no-sub --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('order',$v,'batch 1', d['value'], d['ms_per_step'], d.get('parity_l1'), [(k['key'], round(k['avg_launch_ms']*1e3,1)) for k in d['roofline_kernels']])" | tee -a $OUT/order_ab.txt
done
done
