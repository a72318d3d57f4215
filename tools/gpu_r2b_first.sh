#!/bin/bash
# round 2, second half, first GPU call: the tests touching the changed kernels, the per-kernel A/B, whole-sample A/B at B=1 / B=32
OUT=gpurun_out/${1:-r2b1}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops_f16_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "attention or resid or qkv or epilogue" > $OUT/pytest_ops.txt 2>&1; tail -5 $OUT/pytest_ops.txt
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_reference_golden_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "not rccl and not from_pretrained" > $OUT/pytest_model.txt 2>&1; tail -8 $OUT/pytest_model.txt
timeout 600 python tools/r2b_ab.py f16 > $OUT/kernel_ab.txt 2>&1; cat $OUT/kernel_ab.txt
for arms in "" "gflags=8" "attnvar=16 qpremul=0" "gflags=8 attnvar=16 qpremul=0" ""; do
  for b in 1 32; do
    st=4; [ $b = 32 ] && st=2
    echo -n "B=$b [$arms] " | tee -a $OUT/sample_ab.txt
    timeout 600 python tools/bench_flags.py $arms -- --batch $b --steps $st --warmup 1 --no-cpu-baseline --no-sub 2>$OUT/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms  parity_l1', d.get('parity_l1'))" | tee -a $OUT/sample_ab.txt
  done
done
