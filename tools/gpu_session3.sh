#!/bin/bash
TAG=${1:-s3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; tail -3 $OUT/pytest.txt
timeout 300 python tools/microbench.py > $OUT/microbench.txt 2>&1; cat $OUT/microbench.txt
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/bench_b1.json 2> $OUT/bench_b1.err; cut -c1-250 $OUT/bench_b1.json
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_b32 -o b32 -- python $R/bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline > $R/$OUT/prof_b32.json 2> $R/$OUT/prof_b32.err
cd $R
cut -c1-250 $OUT/prof_b32.json
find $OUT/prof_b32 -type f | head; find $OUT/prof_b32 -name "*kernel_trace.csv" -size +30M -delete
