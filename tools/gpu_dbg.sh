#!/bin/bash
OUT=gpurun_out/${1:-r2d}
mkdir -p $OUT
timeout 600 python tools/attn_prio_bench.py 64 > $OUT/attn_v5.txt 2>&1; cat $OUT/attn_v5.txt
