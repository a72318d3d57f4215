#!/bin/bash
# round 6: fragment reads of the ring kernels double-buffered per K sub-step -- GEMM op tests and model parity, then previous library vs
# this one at sample() level in alternating processes (batch 1, 2, 4, 8: the sizes the ring kernels serve)
TAG=${1:-r6f}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_ops_gpu.py tests/test_ops_f16_gpu.py -q -m gpu -x -p no:cacheprovider > $OUT/t_ops.log 2>&1; echo "ops rc=$?"; tail -2 $OUT/t_ops.log
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_reference_golden_gpu.py -q -m gpu -x -p no:cacheprovider -k "parity or golden or mid_batches or ln_fold or graph" > $OUT/t_model.log 2>&1; echo "model rc=$?"; tail -2 $OUT/t_model.log
R=$GRAFT_REPO_ROOT
for rep in 1 2 3 4; do
  for lib in libf5tts_hip_prev.so libf5tts_hip.so; do
    F5_AB_BATCHES=1,2,4,8 F5_AB_REPS=6 F5TTS_HIP_LIB=$R/f5_tts_mlx_amd/csrc/$lib timeout 600 python tools/experiments/lib_ab_sample.py 2>/dev/null | tail -1 | tee -a $OUT/lib_ab_ring_frag_prefetch.jsonl
  done
done
