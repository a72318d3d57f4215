#!/bin/bash
# round 2, second half, second GPU call: residual-update tests, preload / prefetch A/B per kernel and per sample
OUT=gpurun_out/${1:-r2c1}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_ops_f16_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "resid" > $OUT/pytest_ops.txt 2>&1; tail -5 $OUT/pytest_ops.txt
timeout 600 python tools/r2c_ab.py f16 > $OUT/kernel_ab.txt 2>&1; cat $OUT/kernel_ab.txt
for arms in "" "gflags=256" "" "gflags=256"; do
    echo -n "B=1 [$arms] " | tee -a $OUT/sample_ab.txt
    timeout 600 python tools/bench_flags.py $arms -- --batch 1 --steps 5 --warmup 2 --no-cpu-baseline --no-sub 2>$OUT/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms  parity_l1', d.get('parity_l1'))" | tee -a $OUT/sample_ab.txt
done
for arms in "" "gflags=2048" "gflags=1024" "" "gflags=2048"; do
    echo -n "B=32 [$arms] " | tee -a $OUT/sample_ab.txt
    timeout 600 python tools/bench_flags.py $arms -- --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-sub 2>$OUT/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms  parity_l1', d.get('parity_l1'))" | tee -a $OUT/sample_ab.txt
done
