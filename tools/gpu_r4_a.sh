#!/bin/bash
# round-4 call A: the new / changed tests, the yardstick (vendor GEMM + vendor attention + MFMA loop next to our kernels), sample-level
# A/B of the in-wave pipelined attention kernel
TAG=${1:-r4a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_ops_gpu.py tests/test_ops_f16_gpu.py -m gpu -q -rA --tb=short -p no:cacheprovider -k "v2p or rs128_several or attention_without_tile_maximum or resid_gate" > $OUT/pytest_ops.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_ops.txt; grep -E "passed|failed" $OUT/pytest_ops.txt | tail -2
timeout 600 python tools/yardstick.py mfma attention gemm > $OUT/yardstick.jsonl 2> $OUT/yardstick.err; cut -c1-400 $OUT/yardstick.jsonl
timeout 600 python tools/sample_ab.py --batch 32 --rounds 3 --iters 2 attn_pipe=0 attn_pipe=1 > $OUT/sample_ab_b32.json 2> $OUT/sample_ab_b32.err; cat $OUT/sample_ab_b32.json
timeout 900 python -m pytest tests/test_model_gpu.py tests/test_reference_golden_gpu.py tests/test_dist_gpu.py -m gpu -q -rA --tb=short -p no:cacheprovider -k "mid_batches or ragged_batch_vs_oracle or benched_f16 or one_utterance or pad_to_respects" > $OUT/pytest_model.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest_model.txt; grep -E "passed|failed|\[mid batch\]|\[ragged|one-utterance" $OUT/pytest_model.txt | tail -16
