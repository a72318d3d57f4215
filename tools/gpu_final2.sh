#!/bin/bash
# end-of-round check of the final build: full GPU tests, smoke, bench B=1 / B=32 (no CPU baseline leg)
TAG=${1:-final2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; tail -3 $OUT/pytest.txt
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 300 python bench.py --no-cpu-baseline > $OUT/bench_b1.json 2> $OUT/bench_b1.err; cut -c1-160 $OUT/bench_b1.json
timeout 300 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_b32.json 2> $OUT/bench_b32.err; cut -c1-160 $OUT/bench_b32.json
