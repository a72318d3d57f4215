#!/bin/bash
TAG=${1:-s2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; tail -3 $OUT/pytest.txt
timeout 300 python tools/microbench.py > $OUT/microbench.txt 2>&1; cat $OUT/microbench.txt
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_b1 -o b1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/$OUT/prof_b1.json 2> $GRAFT_REPO_ROOT/$OUT/prof_b1.err
cd $GRAFT_REPO_ROOT
timeout 600 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_b32.json 2> $OUT/bench_b32.err; cut -c1-400 $OUT/bench_b32.json
timeout 600 python bench.py --batch 1 --precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_b1_x3.json 2> $OUT/bench_b1_x3.err; cut -c1-300 $OUT/bench_b1_x3.json
ls -la $OUT/prof_b1* | head; find $OUT/prof_b1 -name "*stats*" | head
# keep only the small csv summaries (the per-dispatch trace can be large)
find $OUT/prof_b1 -name "*kernel_trace.csv" -size +20M -delete
