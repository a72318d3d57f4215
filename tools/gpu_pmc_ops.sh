#!/bin/bash
# HBM-traffic passes over the five block kernels: tools/gpu_pmc_ops.sh TAG
OUT=gpurun_out/${1:-pmcops}
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for B in 1 32; do
  D=$R/$OUT/b${B}_f16
  mkdir -p $D
  cd /tmp
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $D/fetch -o pmc -- python $R/tools/pmc_block_ops.py f16 $B > $D/driver.txt 2> $D/fetch.err
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $D/write -o pmc -- python $R/tools/pmc_block_ops.py f16 $B > /dev/null 2> $D/write.err
  cd $R
  python tools/pmc_block_ops_summary.py $D f16 $B > $OUT/traffic_b${B}_f16.json 2> $OUT/traffic_b${B}_f16.err
  find $D -name "*kernel_trace.csv" -delete; find $D -name "*.csv" -size +4M -delete
  cat $OUT/traffic_b${B}_f16.json | head -80; tail -3 $OUT/traffic_b${B}_f16.err
done
