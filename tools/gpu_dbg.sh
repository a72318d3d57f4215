#!/bin/bash
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_ops_f16_gpu.py -m gpu -q --tb=short -p no:cacheprovider -k "gemm or qkv" > $OUT/pytest_gemm.txt 2>&1; grep -E "passed|failed" $OUT/pytest_gemm.txt | tail -2; grep -E "^(FAILED|ERROR)" $OUT/pytest_gemm.txt | head -20
