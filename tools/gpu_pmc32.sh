#!/bin/bash
# PMC passes at B=32 (eager, one sample): MFMA busy, LDS activity / conflicts, wave wait states per kernel
OUT=gpurun_out/${1:-pmc32}
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for C in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_INSTS_MFMA"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$OUT/pmc_$N -o pmc -- python $R/bench.py --batch 32 --steps 1 --warmup 0 --no-cpu-baseline --no-graph --ode-points 3 > $R/$OUT/pmc_$N.json 2> $R/$OUT/pmc_$N.err
done
cd $R
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
cut -c1-330 $OUT/pmc_summary.txt | head -40
