#!/bin/bash
# round 6, after the last code change: the full GPU suite, smoke and the default bench line once more (no profiler passes)
TAG=${1:-r6confirm}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; grep -E "passed|failed" $OUT/pytest.txt | tail -2; grep -E "^FAILED" $OUT/pytest.txt | head
grep -E "^\[(ln_fold|f16 range|f16 saturation|bf16 \+ ln_fold|batch32|batch 32 vs oracle|full-size|ragged|mid batch|one-utterance|batch_sentences)" $OUT/pytest.txt | sort -u > $OUT/gpu_tests_printed_figures.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -1 $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-400 $OUT/bench_default.json
