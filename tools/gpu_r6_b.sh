#!/bin/bash
# round-6 call B: the fp16 range detector -- its tests, then its cost (previous library vs this one, alternating processes)
TAG=${1:-r6b}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_ops_f16_gpu.py -q -m gpu -x -k "range_detector or saturate or ln_modulate_f16 or epilogues_f16" -s > $OUT/t_ops.log 2>&1; echo "ops rc=$?"; tail -3 $OUT/t_ops.log
timeout 2400 python -m pytest tests/test_model_gpu.py -q -m gpu -x -s -k "range_stress_outlier or overflow_falls_back or auto_range_check or batch32_every_utterance_vs_oracle or smoke or sample_parity_f16" > $OUT/t_model.log 2>&1; echo "model rc=$?"; tail -5 $OUT/t_model.log; grep "f16 range\|batch 32 vs oracle" $OUT/t_model.log
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in libf5tts_hip_prev.so libf5tts_hip.so; do
    F5_AB_BATCHES=1,32 F5_AB_REPS=4 F5TTS_HIP_LIB=$R/f5_tts_mlx_amd/csrc/$lib timeout 600 python tools/experiments/lib_ab_sample.py 2>/dev/null | tail -1 | tee -a $OUT/lib_ab_sat_detector.jsonl
  done
done
