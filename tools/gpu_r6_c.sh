#!/bin/bash
# round-6 call C: group-major rotation tables (one 16-byte load per lane and 4 features, scalar-base addressing) -- tests of every QKV
# path, then previous library vs this one at sample() level (alternating processes) and at op level (yardstick, this library only)
TAG=${1:-r6c}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout 2400 python -m pytest tests/test_ops_gpu.py tests/test_ops_f16_gpu.py -q -m gpu -x -k "qkv or attention or rs128 or folded or range_detector" > $OUT/t_ops.log 2>&1; echo "ops rc=$?"; tail -2 $OUT/t_ops.log
timeout 2400 python -m pytest tests/test_model_gpu.py tests/test_reference_golden_gpu.py -q -m gpu -x -k "parity or golden or mid_batches or batch32_every_utterance_vs_oracle or graph_split" > $OUT/t_model.log 2>&1; echo "model rc=$?"; tail -2 $OUT/t_model.log
R=$GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in libf5tts_hip_prev.so libf5tts_hip.so; do
    F5_AB_BATCHES=1,8,32 F5_AB_REPS=4 F5TTS_HIP_LIB=$R/f5_tts_mlx_amd/csrc/$lib timeout 600 python tools/experiments/lib_ab_sample.py 2>/dev/null | tail -1 | tee -a $OUT/lib_ab_g4_tables.jsonl
  done
done
timeout 300 python tools/yardstick.py gemm 2>/dev/null | grep '"shape": "qkv"' | grep workload | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(json.dumps({'shape': r['shape'], 'us': r['us'], 'tflops': r['tflops']}))" | tee $OUT/yardstick_qkv.jsonl
