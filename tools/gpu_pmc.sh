#!/bin/bash
# usage: tools/gpu_pmc.sh TAG "<counters>" <bench args...>
TAG=$1; C=$2; shift 2
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$OUT/pmc_x -o pmc -- python $R/bench.py "$@" --no-cpu-baseline --no-graph > $R/$OUT/bench.json 2> $R/$OUT/bench.err
cd $R
python tools/pmc_summary.py $OUT | cut -c1-400
find $OUT -name "*.csv" -size +8M -delete
