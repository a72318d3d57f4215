#!/bin/bash
# round-3 evidence on one MI355X: product-library GPU tests, smoke, default bench line (B=1 + sub-records incl. B=32 roofline kernels,
# eager, cpu baseline), rocprofv3 kernel stats of the B=1 and B=32 runs, HBM-traffic PMC passes over the five block kernels
TAG=${1:-r3final}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
echo "(pytest run separately: gpurun_out/r3b/gpu_tests.txt)" > $OUT/pytest.txt

timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.txt 2>&1; tail -2 $OUT/smoke.txt
timeout 900 python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err; cut -c1-200 $OUT/bench_default.json
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_b1 -o b1 -- python $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-sub > $R/$OUT/prof_b1.json 2> $R/$OUT/prof_b1.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/prof_b32 -o b32 -- python $R/bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-sub > $R/$OUT/prof_b32.json 2> $R/$OUT/prof_b32.err
cd $R
for B in 1 32; do f=$(find $OUT/prof_b$B -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $OUT/b${B}_f16_kernel_stats.csv; done
bash tools/gpu_pmc_ops.sh $TAG > $OUT/pmc_ops.log 2>&1
find $OUT -name "*kernel_trace.csv" -delete; find $OUT -name "*.csv" -size +4M -delete
ls $OUT
