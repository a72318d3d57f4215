#!/bin/bash
TAG=${1:-s4}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; tail -8 $OUT/pytest.txt
cd /tmp
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$OUT/pmc_$N -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $R/$OUT/pmc_$N.json 2> $R/$OUT/pmc_$N.err
  echo "pmc $N exit $?"
done
cd $R
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1; cat $OUT/pmc_summary.txt
find $OUT -name "*.csv" -size +8M -delete
