#!/bin/bash
# second PMC round at B=32 (eager, 3-point solve): per-unit active cycles, LDS waits, L2 hit rate, TA stalls
OUT=gpurun_out/${1:-pmc32b}
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
for C in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS" "SQ_INST_CYCLES_SALU SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCP_TCC_READ_REQ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCP_TA_ADDR_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum" "SQ_LDS_ADDR_CONFLICT SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL SQ_LDS_UNALIGNED_STALL SQ_IFETCH SQ_WAVES"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$OUT/pmc_$N -o pmc -- python $R/bench.py --batch 32 --steps 1 --warmup 0 --no-cpu-baseline --no-graph --ode-points 3 > $R/$OUT/pmc_$N.json 2> $R/$OUT/pmc_$N.err
done
cd $R
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
grep -E "^==|gemm256_kernel<5|gemm256_kernel<4|attn2w" $OUT/pmc_summary.txt | cut -c1-420
