#!/bin/bash
# round 2, second GPU call: the whole GPU suite (no -x), smoke(), default bench line, rocprofv3 kernel stats of B=1 and B=32 (f16)
OUT=gpurun_out/${1:-r2b}
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -rA > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; tail -8 $OUT/pytest.txt; grep -E "^(FAILED|ERROR)" $OUT/pytest.txt | head -30
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json; tail -3 $OUT/bench_default.err
for b in 1 32; do
  st=3; [ $b = 32 ] && st=2
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_b$b -o b$b -- python $GRAFT_REPO_ROOT/bench.py --batch $b --steps $st --warmup 1 --no-cpu-baseline --no-sub > $GRAFT_REPO_ROOT/$OUT/prof_b$b.log 2>&1)
  f=$(find $OUT/prof_b$b -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $OUT/b${b}_f16_kernel_stats.csv && head -9 "$f" | cut -c1-150
  rm -rf $OUT/prof_b$b
done
