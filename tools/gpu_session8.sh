#!/bin/bash
TAG=${1:-s8}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; tail -6 $OUT/pytest.txt
timeout 300 python tools/microbench.py > $OUT/microbench.txt 2>&1; grep -E "attention|attn version" $OUT/microbench.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_b1.json 2> $OUT/bench_b1.err; cut -c1-200 $OUT/bench_b1.json
timeout 600 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_b32.json 2> $OUT/bench_b32.err; cut -c1-200 $OUT/bench_b32.json
