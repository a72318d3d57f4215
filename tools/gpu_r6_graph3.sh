#!/bin/bash
# round 6: does the packet-capture default set by the package (after `import torch`, before the first HIP call) take effect?
# tools/r6_graph_probe.py (alternating legs, medians) with nothing exported, with 0 exported and with 1 exported, twice
TAG=${1:-r6g3}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
run() { env "$@" timeout 300 python tools/r6_graph_probe.py 1 2>/dev/null | tail -1 | python -c "
import sys, json
r = json.loads(sys.stdin.read()); print(json.dumps({'env': '$*', 'eager': r['eager']['ms_median'], 'graph': r['graph']['ms_median'], 'graph_minus_eager': round(r['graph']['ms_median'] - r['eager']['ms_median'], 3)}))" | tee -a $OUT/graph_env_effect.jsonl; }
for rnd in 1 2; do
  run NOTHING_EXPORTED=1
  run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
  run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
done
