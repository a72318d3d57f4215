#!/bin/bash
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
timeout 1800 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest_all.txt 2>&1; grep -E "passed|failed" $OUT/pytest_all.txt | tail -2; grep -E "^(FAILED|ERROR)" $OUT/pytest_all.txt | head -20
