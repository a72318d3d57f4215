#!/bin/bash
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_model_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "fused_ln or resid or ln_modulate" > $OUT/pytest_ln.txt 2>&1; grep -E "passed|failed" $OUT/pytest_ln.txt | tail -2; grep -E "^(FAILED|ERROR)|Error|assert" $OUT/pytest_ln.txt | head -20
for rep in 1 2; do
for v in 1 0; do
  timeout 300 python tools/bench_flags.py lnfuse=$v -- --batch 1 --steps 4 --warmup 2 --no-sub --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('lnfuse',$v, d['value'], d['ms_per_step'], d.get('parity_l1'))" | tee -a $OUT/lnfuse_ab.txt
done
done
