#!/bin/bash
TAG=${1:-s5}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -q -rA --tb=short -p no:cacheprovider -k "v2 or gemm" > $OUT/pytest_v2.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_v2.txt; grep -E "parity\] gemm256|passed|failed|FAILED" $OUT/pytest_v2.txt | tail -30
timeout 300 python tools/microbench.py > $OUT/microbench.txt 2>&1; grep -E "v2|M\": 59968" $OUT/microbench.txt
