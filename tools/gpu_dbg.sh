#!/bin/bash
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops_f16_gpu.py -m gpu -q --tb=short -p no:cacheprovider -x -k "convpos or fused_ln" > $OUT/pytest_cp.txt 2>&1; grep -E "passed|failed" $OUT/pytest_cp.txt | tail -2; grep -E "^(FAILED|ERROR)|Error|assert" $OUT/pytest_cp.txt | head -20
timeout 600 python tools/convpos_bench.py > $OUT/convpos_bench.txt 2>&1; cat $OUT/convpos_bench.txt
