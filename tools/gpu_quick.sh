#!/bin/bash
# quick regression + perf check: full GPU tests, bench B=1 / B=32 (no CPU baseline)
OUT=gpurun_out/${1:-quick}
mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; tail -4 $OUT/pytest.txt
timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_b1.json 2> $OUT/bench_b1.err; cut -c1-200 $OUT/bench_b1.json
timeout 600 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_b32.json 2> $OUT/bench_b32.err; cut -c1-200 $OUT/bench_b32.json
