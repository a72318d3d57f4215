#!/bin/bash
# round-4 call C: LN fold -- op / model parity tests, then the A/B on sample()
TAG=${1:-r4fold}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops_f16_gpu.py -m gpu -q -rA --tb=short -p no:cacheprovider -k "ln_modulate_folded or ln_fold" > $OUT/pytest_ops.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_ops.txt; grep -E "passed|failed" $OUT/pytest_ops.txt | tail -2; grep -E "^FAILED|^E  " $OUT/pytest_ops.txt | head -30; grep "LN fold" $OUT/pytest_ops.txt | head -40
timeout 600 python tools/r4_ln_fold_ab.py --batches ${BATCHES:-4,8,32} --reps 2 > $OUT/ln_fold_ab.jsonl 2> $OUT/ab_err.txt; cat $OUT/ln_fold_ab.jsonl; tail -3 $OUT/ab_err.txt
timeout 1200 python -m pytest tests/test_model_gpu.py -m gpu -q -rA --tb=short -p no:cacheprovider -k "ln_fold" > $OUT/pytest_model.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_model.txt; grep -E "passed|failed" $OUT/pytest_model.txt | tail -2; grep -E "^FAILED|^E  |ln_fold\]" $OUT/pytest_model.txt | head -30
