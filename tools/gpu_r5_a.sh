#!/bin/bash
# round-5 call A: epilogue ablations / phase shift / band width on the measurement build, and the vendor GEMM's anatomy
TAG=${1:-r5a}
OUT=gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
F5TTS_HIP_LIB=$R/f5_tts_mlx_amd/csrc/libf5tts_hip_probe.so timeout 600 python tools/r5_epilogue_probe.py > $OUT/epilogue_probe.jsonl 2> $OUT/epilogue_probe.err
tail -3 $OUT/epilogue_probe.err; python - <<P
import json
for l in open("$OUT/epilogue_probe.jsonl"):
    r = json.loads(l); print(r["shape"], json.dumps(r["min_us"]))
P
cd /tmp
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/va_trace -o va -- python $R/tools/r5_vendor_anatomy.py run > $R/$OUT/va_trace.log 2>&1
timeout 300 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$OUT/va_pmc -o va -- python $R/tools/r5_vendor_anatomy.py run > $R/$OUT/va_pmc.log 2>&1
cd $R
python tools/r5_vendor_anatomy.py summarize $OUT/va_trace > $OUT/vendor_gemm_anatomy_trace.jsonl 2> $OUT/va_sum.err
python tools/r5_vendor_anatomy.py summarize $OUT/va_pmc > $OUT/vendor_gemm_anatomy_pmc.jsonl 2>> $OUT/va_sum.err
for f in $(find $OUT/va_trace -name "*kernel_trace.csv" | head -1); do head -1 $f > $OUT/kernel_trace_header.txt; grep -i "cijk\|gemm\|attn\|fmha\|flash" $f | awk -F, '!seen[$8]++' | head -40 > $OUT/kernel_trace_first_rows.csv; done
cut -c1-700 $OUT/vendor_gemm_anatomy_trace.jsonl | head -20
find $OUT -name "*.csv" -size +2M -delete
ls $OUT
