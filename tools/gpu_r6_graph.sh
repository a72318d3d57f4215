#!/bin/bash
# round-6: hipGraph replay vs eager at batch 1 under the runtime's graph knobs (environment variables of libamdhip64, read at start-up)
TAG=${1:-r6g}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { echo "== $*"; env "$@" timeout 300 python tools/r6_graph_probe.py 1 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print(json.dumps({'env': '$*', 'eager': r['eager']['ms_median'], 'graph': r['graph']['ms_median'], 'graph_split': r['graph_split']['ms_median'], 'bit_identical': r['bit_identical'], 'first_graph_call_ms': r['graph']['first_call_ms']}))" | tee -a $OUT/graph_knobs.jsonl; }
run X=0
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run DEBUG_HIP_GRAPH_BATCH_SIZE=256
run DEBUG_HIP_GRAPH_BATCH_SIZE=8192
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_MAX_BATCH_SIZE=4096
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
