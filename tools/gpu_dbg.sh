#!/bin/bash
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_model_gpu.py tests/test_reference_golden_gpu.py -m gpu -q --tb=short -p no:cacheprovider -s -k "noise or seed or raw_wave or generate or sample" > $OUT/pytest_noise.txt 2>&1; grep -E "passed|failed|\[noise\]" $OUT/pytest_noise.txt | tail -14; grep -E "^(FAILED|ERROR)|Error|assert " $OUT/pytest_noise.txt | head
