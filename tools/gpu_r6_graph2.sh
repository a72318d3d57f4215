#!/bin/bash
# round 6: the bench's own headline (hipGraph replay, per-call status read) vs --no-graph, with the package's packet-capture default and
# with the runtime's default forced back on; two rounds in alternation
TAG=${1:-r6g2}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
one() { name=$1; extra=$2; shift 2; env "$@" timeout 300 python bench.py --steps 12 --warmup 3 --no-sub --no-cpu-baseline $extra 2>/dev/null | python -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1]); print(json.dumps({'case': '$name', 'ms_per_step': round(r['ms_per_step'], 3)}))" | tee -a $OUT/bench_graph_vs_eager.jsonl; }
for rnd in 1 2; do
  one "graph, package default (capture off)" "" X=0
  one "eager (--no-graph)" --no-graph X=0
  one "graph, DEBUG_CLR_GRAPH_PACKET_CAPTURE=1 exported" "" DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
  one "graph, DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 exported" "" DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
done
