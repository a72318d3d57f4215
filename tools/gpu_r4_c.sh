#!/bin/bash
# round-4 call: band-major ring numbering + batched fold constants -- tests, batch-1 / batch-32 bench lines, A/B of the numbering
TAG=${1:-r4c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1200 python -m pytest tests/test_ops_gpu.py tests/test_ops_f16_gpu.py tests/test_model_gpu.py -m gpu -q -rA --tb=short -p no:cacheprovider -k "band_major or ln_fold or ln_modulate_folded or ring8 or resid_gate or small_tile" > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; grep -E "passed|failed" $OUT/pytest.txt | tail -2; grep -E "^FAILED|^E  " $OUT/pytest.txt | head -20
timeout 600 python tools/r4_ring_order_ab.py > $OUT/ring_order_ab.json 2> $OUT/ring_order_ab.err; cat $OUT/ring_order_ab.json; tail -2 $OUT/ring_order_ab.err
timeout 600 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-sub > $OUT/bench_b32.json 2> $OUT/bench_b32.err
python -c "import json;d=json.load(open('$OUT/bench_b32.json'));print('b32', d['ms_per_step'])"
