#!/bin/bash
# round 6: (1) us per launch of the four block GEMMs against M across the 256-CU round boundaries, (2) two half batches on two streams
TAG=${1:-r6d}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 600 python tools/r6_m_sweep.py > $OUT/m_sweep.jsonl 2> $OUT/m_sweep.err; tail -3 $OUT/m_sweep.err
python - <<PY
import json
rows=[json.loads(l) for l in open("$OUT/m_sweep.jsonl")]
for s in ("qkv","out_proj","ff1","ff2"):
    print(s)
    for r in rows:
        if r["shape"]==s and r["round"]==1: print("  b %2d tiles %5d rounds %6.2f us %7.1f  us/round %6.2f  TF %5d" % (r["batch"], r["tiles"], r["rounds_of_256"], r["us"], r["us_per_tile_round"], r["tflops"]))
PY
timeout 900 python tools/r6_two_stream_probe.py 3 > $OUT/two_streams.jsonl 2> $OUT/two_streams.err; tail -3 $OUT/two_streams.err; cat $OUT/two_streams.jsonl
