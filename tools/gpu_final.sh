#!/bin/bash
# final round evidence: full GPU tests, smoke, bench B=1 (with cpu baseline) / B=32 / bf16x3, rocprof kernel stats, PMC passes
TAG=${1:-final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > $OUT/pytest.txt 2>&1
echo "pytest exit $?" >> $OUT/pytest.txt; tail -3 $OUT/pytest.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 600 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-160 $OUT/bench_default.json
timeout 600 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_b32.json 2> $OUT/bench_b32.err; cut -c1-160 $OUT/bench_b32.json
timeout 600 python bench.py --precision bf16x3 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_b1_x3.json 2> $OUT/bench_b1_x3.err; cut -c1-160 $OUT/bench_b1_x3.json
timeout 600 python bench.py --precision bf16x3 --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $OUT/bench_b32_x3.json 2> $OUT/bench_b32_x3.err; cut -c1-160 $OUT/bench_b32_x3.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_b1 -o b1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $R/$OUT/prof_b1.json 2> $R/$OUT/prof_b1.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_b32 -o b32 -- python $R/bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $R/$OUT/prof_b32.json 2> $R/$OUT/prof_b32.err
for C in FETCH_SIZE WRITE_SIZE "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"; do
  N=$(echo $C | cut -d' ' -f1)
  timeout 600 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $R/$OUT/pmc_$N -o pmc -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-graph > $R/$OUT/pmc_$N.json 2> $R/$OUT/pmc_$N.err
done
cd $R
python tools/pmc_summary.py $OUT > $OUT/pmc_summary.txt 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*counter_collection.csv" -size +8M -delete
ls $OUT
