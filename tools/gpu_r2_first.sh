#!/bin/bash
# round 2, first GPU call: full GPU tests of the fp16-operand build, the default bench line (f16, sub-records), rocprof of B=1,
# B=32 line, per-XCD bandwidth probe
OUT=gpurun_out/${1:-r2a}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x --tb=short -p no:cacheprovider -rA > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; tail -5 $OUT/pytest.txt
grep "\[parity\].*\(f16\|full-length\)" $OUT/pytest.txt | tail -40
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-400 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
timeout 600 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-sub > $OUT/bench_b32.json 2> $OUT/bench_b32.err; cut -c1-300 $OUT/bench_b32.json
timeout 120 tools/probes/xcdbw.bin > $OUT/xcdbw.txt 2>&1; cat $OUT/xcdbw.txt
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$OUT/prof_b1 -o b1 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-sub > $GRAFT_REPO_ROOT/$OUT/prof_b1.log 2>&1)
find $OUT/prof_b1 -name "*kernel_stats*" | head; f=$(find $OUT/prof_b1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -8 "$f" | cut -c1-160
# keep the merged output small: drop the raw traces
find $OUT/prof_b1 -name "*kernel_trace*" -delete 2>/dev/null
