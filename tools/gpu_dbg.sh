#!/bin/bash
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $OUT/pytest.txt 2>&1; tail -4 $OUT/pytest.txt; grep -E "^(FAILED|ERROR)" $OUT/pytest.txt | head
for i in 1 2 3; do timeout 600 python bench.py --no-cpu-baseline > $OUT/bench_$i.json 2> $OUT/bench_$i.err; python -c "
import json,sys; d=json.load(open('$OUT/bench_$i.json')); print('bench $i', round(d['ms_per_step'],2), d.get('parity_l1'), d['sub']['b32_f16']['ms_per_step'], d['sub']['b32_f16'].get('parity_l1'), d['sub']['b1_bf16']['ms_per_step'], d['sub']['wave_to_wave']['ms_per_step'])"; done
