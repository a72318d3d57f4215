#!/bin/bash
# round-6 call A: the one-wave-per-SIMD / two-accumulator-set probe (tools/probes/onewave_gemm.hip) next to the shipped kernels' own
# main-loop and full-launch numbers on the same box (tools/yardstick.py gemm mfma)
TAG=${1:-r6a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
for N in 1024 2048; do
  timeout 300 tools/probes/bin/onewave_gemm $N 9 20 > $OUT/onewave_N$N.jsonl 2> $OUT/onewave_N$N.err
  echo "rc=$? N=$N"; cat $OUT/onewave_N$N.jsonl; tail -3 $OUT/onewave_N$N.err
done
[ -n "$SKIP_YARD" ] || timeout 600 python tools/yardstick.py gemm mfma > $OUT/yardstick.jsonl 2> $OUT/yardstick.err
tail -2 $OUT/yardstick.err
python - <<P
import json
for l in open("$OUT/yardstick.jsonl"):
    try: r = json.loads(l)
    except Exception: continue
    print(json.dumps(r)[:400])
P
