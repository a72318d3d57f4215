#!/bin/bash
# One GPU-box session: unit + model parity tests, bench, rocprof kernel trace. Everything lands in gpurun_out/.
# usage: tools/gpu_session.sh [tag]
TAG=${1:-s}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
echo "== rocminfo ==" > $OUT/env.txt; (rocminfo | grep -E "gfx|Compute Unit|Marketing" | head -8; nproc; lscpu | grep "Model name") >> $OUT/env.txt 2>&1
timeout 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt
tail -5 $OUT/pytest.txt
timeout 600 python bench.py --steps 5 --warmup 2 > $OUT/bench_b1.json 2> $OUT/bench_b1.err
echo "bench exit $?"; cat $OUT/bench_b1.json | cut -c1-600
