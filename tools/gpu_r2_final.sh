#!/bin/bash
# round 2, evidence call on the final build: whole GPU suite, smoke(), default bench line, rocprofv3 kernel stats (B=1, B=32, f16),
# FETCH_SIZE / WRITE_SIZE passes over the five block kernels
OUT=gpurun_out/${1:-r2final}
bash tools/gpu_r2_second.sh ${1:-r2final}
bash tools/gpu_pmc_ops.sh ${1:-r2final} > $OUT/pmc_ops.log 2>&1
grep -E "\"key\"|fetch_bytes|write_bytes\"" $OUT/traffic_b32_f16.json | head -20
