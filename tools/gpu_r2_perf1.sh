#!/bin/bash
# round 2 perf experiments 1: attention issue-priority schemes, 128x256 two-per-CU GEMM with epilogue priority
OUT=gpurun_out/${1:-r2c}
mkdir -p $OUT
timeout 300 python -m pytest tests/test_model_gpu.py -m gpu -q -x --tb=short -p no:cacheprovider -k "rccl or vocoder_c_abi" > $OUT/pytest_sel.txt 2>&1; tail -3 $OUT/pytest_sel.txt
timeout 600 python tools/attn_prio_bench.py 64 8 > $OUT/attn_prio.txt 2>&1; cat $OUT/attn_prio.txt
timeout 900 python tools/gemm_v3_prio_bench.py f16 > $OUT/gemm_v3_prio_f16.txt 2>&1; cat $OUT/gemm_v3_prio_f16.txt
timeout 600 python tools/gemm_v3_prio_bench.py bf16 > $OUT/gemm_v3_prio_bf16.txt 2>&1; cat $OUT/gemm_v3_prio_bf16.txt
