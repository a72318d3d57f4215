#!/bin/bash
# round 2: transposed-tile 16-bit epilogues of the 256x256 kernel (FF1, q / k of QKV) -- tests, kernel A/B, sample A/B at batch 32
OUT=gpurun_out/${1:-r2e2}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py tests/test_ops_f16_gpu.py tests/test_mxfp8_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -s -k "gemm or qkv or attention_path" > $OUT/pytest_ops.txt 2>&1; grep "transposed vs straight" $OUT/pytest_ops.txt | head -12; tail -4 $OUT/pytest_ops.txt
timeout 600 python -m pytest tests/test_model_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "full_size or batch or parity" > $OUT/pytest_model.txt 2>&1; tail -4 $OUT/pytest_model.txt
timeout 600 python tools/tr_epilogue_ab.py f16 > $OUT/kernel_ab.txt 2>&1; cat $OUT/kernel_ab.txt
for arms in "" "qkvtr=0" "gflags=16384 qkvtr=0" "" "qkvtr=0"; do
    echo -n "B=32 [$arms] " | tee -a $OUT/sample_ab.txt
    timeout 600 python tools/bench_flags.py $arms -- --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-sub 2>$OUT/ab.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],2), 'ms  parity_l1', d.get('parity_l1'))" | tee -a $OUT/sample_ab.txt
done
