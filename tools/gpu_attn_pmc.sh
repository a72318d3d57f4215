#!/bin/bash
# wave-time decomposition (PMC) of the attention kernels at the B=32 shape: round-1 kernel (2) vs in-wave pipelined (5)
OUT=gpurun_out/${1:-attnpmc}
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for V in 2 5; do
  i=0
  for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_MFMA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_TRANS GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$OUT/pmc_v${V}_$i -o pmc -- python $R/tools/attn_pmc_driver.py $V f16 > $R/$OUT/pmc_v${V}_$i.log 2>&1
  done
done
cd $R
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete
grep -E "==|attn" $OUT/pmc_summary.txt | cut -c1-600
