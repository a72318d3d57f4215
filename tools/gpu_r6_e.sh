#!/bin/bash
# round 6: the split sample() -- its tests, then unsplit vs split by batch size
TAG=${1:-r6e}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_model_gpu.py -m gpu -q -x -k "split_sample or batch32_every_utterance_vs_oracle or saturation_detector or two_engines or graph_cache" -p no:cacheprovider > $OUT/pytest.txt 2>&1; tail -15 $OUT/pytest.txt
timeout 1200 python tools/r6_split_ab.py 16 32 64 > $OUT/split_ab.jsonl 2> $OUT/split_ab.err; tail -3 $OUT/split_ab.err; cut -c1-700 $OUT/split_ab.jsonl
