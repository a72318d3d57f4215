#!/bin/bash
# round-6 evidence on one MI355X: product-library GPU tests, smoke, default bench line (B=1 + sub-records incl. B=32 roofline kernels,
# measured MFMA peak, eager, cpu baseline), BASELINE configs[4] (c5), rocprofv3 kernel stats of the B=1 and B=32 runs, PMC passes over the
# five block kernels: HBM traffic (FETCH_SIZE / WRITE_SIZE) and MFMA utilisation (SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE)
TAG=${1:-r6final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 2700 python -m pytest tests -m gpu -q -rA --tb=short -p no:cacheprovider > $OUT/pytest.txt 2>&1
grep -E "^\[(ln_fold|f16 range|f16 saturation|bf16 \+ ln_fold|batch32|batch 32 vs oracle|full-size|ragged|mid batch|one-utterance|batch_sentences)" $OUT/pytest.txt | sort -u > $OUT/gpu_tests_printed_figures.txt
echo "pytest exit $?" >> $OUT/pytest.txt; grep -E "passed|failed" $OUT/pytest.txt | tail -2; grep -E "^FAILED" $OUT/pytest.txt | head
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-300 $OUT/bench_default.json
timeout 600 python bench.py --config c5 --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_c5_mxfp8.json 2> $OUT/bench_c5.err; cut -c1-300 $OUT/bench_c5_mxfp8.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_b1 -o b1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub > $R/$OUT/prof_b1.json 2> $R/$OUT/prof_b1.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_b32 -o b32 -- python $R/bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-sub > $R/$OUT/prof_b32.json 2> $R/$OUT/prof_b32.err
cd $R
for B in 1 32; do f=$(find $OUT/prof_b$B -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/b${B}_f16_kernel_stats.csv; done
bash tools/gpu_pmc_ops.sh $TAG > $OUT/pmc_ops.log 2>&1
for B in 1 32; do
  D=$R/$OUT/mfma_b${B}_f16
  mkdir -p $D
  cd /tmp
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F16 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $D -o pmc -- python $R/tools/pmc_block_ops.py f16 $B > $D/driver.txt 2> $D/err.txt
  cd $R
  python tools/pmc_mfma_summary.py $D f16 $B > $OUT/mfma_util_b${B}_f16.json 2> $OUT/mfma_util_b${B}_f16.err
  head -c 1500 $OUT/mfma_util_b${B}_f16.json; tail -2 $OUT/mfma_util_b${B}_f16.err
done
timeout 900 python tools/yardstick.py gemm mfma > $OUT/yardstick_gemm_attention_mfma.jsonl 2> $OUT/yardstick.err; tail -3 $OUT/yardstick_gemm_attention_mfma.jsonl | cut -c1-300
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +4M -delete
ls $OUT
