#!/bin/bash
# round-4 call B: the pipelined attention kernel: parity tests + scaling probe
TAG=${1:-r4b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops_f16_gpu.py -m gpu -q -rA --tb=line -p no:cacheprovider -k "v2p" > $OUT/pytest_ops.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_ops.txt; grep -E "passed|failed" $OUT/pytest_ops.txt | tail -2; grep -E "^FAILED" $OUT/pytest_ops.txt | head -20
timeout 300 python tools/r4_attn_scaling.py > $OUT/attn_scaling.jsonl 2> $OUT/err1.txt; cut -c1-420 $OUT/attn_scaling.jsonl
