#!/bin/bash
# QKV epilogue / 128x256 ring tiles: correctness of everything that goes through the QKV epilogue + QKV timing at batch 1
OUT=gpurun_out/${1:-wide}
mkdir -p $OUT
timeout 150 python -m pytest tests/test_ops_gpu.py -q -x -k "attention or wide_ring or ring8 or qkv" --tb=short -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; tail -3 $OUT/pytest.txt
timeout 120 python tools/qkv_tiles_bench.py > $OUT/qkv_tiles.txt 2>&1; cat $OUT/qkv_tiles.txt | tail -2
