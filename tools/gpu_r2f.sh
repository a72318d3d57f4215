#!/bin/bash
# round 2: batch-1 QKV projection on 8-wave 128x256 ring tiles with transposed q / k wave tiles -- tests, kernel A/B, sample A/B
OUT=gpurun_out/${1:-r2f1}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "qkv" > $OUT/pytest_ops.txt 2>&1; tail -4 $OUT/pytest_ops.txt
timeout 300 python tools/qkv_b1_tiles_ab.py f16 > $OUT/kernel_ab.txt 2>&1; cat $OUT/kernel_ab.txt
for arms in "" "qkvtile=13" "qkvtile=12" "" "qkvtile=13" "qkvtile=12"; do
    echo -n "B=1 [$arms] " | tee -a $OUT/sample_ab.txt
    timeout 300 python tools/bench_flags.py $arms -- --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-sub 2>$OUT/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms  parity_l1', d.get('parity_l1'))" | tee -a $OUT/sample_ab.txt
done
