#!/bin/bash
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops_f16_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "attention" > $OUT/pytest_attn.txt 2>&1; grep -E "passed|failed" $OUT/pytest_attn.txt | tail -2; grep -E "^(FAILED|ERROR)" $OUT/pytest_attn.txt | head -20
timeout 600 python tools/attn_prio_bench.py 64 16 > $OUT/attn_lazy.txt 2>&1; cat $OUT/attn_lazy.txt
