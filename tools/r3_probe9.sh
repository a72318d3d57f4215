#!/bin/bash
# Round-3 probe 9: shader clock and power while the large-grid attention kernel / the 256x256 GEMM loop on workload-like data.
# (A kernel at N % of the MFMA peak at the NOMINAL 2.4 GHz may be at a much higher fraction of what the clock it actually gets allows.)
mkdir -p gpurun_out/r3_probe9
python tools/r3_probe9.py attn > gpurun_out/r3_probe9/attn.log 2>&1 &
PID=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power|mclk|fclk" ; echo "--"; sleep 1; done
wait $PID
tail -2 gpurun_out/r3_probe9/attn.log
python tools/r3_probe9.py gemm > gpurun_out/r3_probe9/gemm.log 2>&1 &
PID=$!
sleep 25
for i in 1 2 3 4 5 6; do rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|Power" ; echo "--"; sleep 1; done
wait $PID
tail -2 gpurun_out/r3_probe9/gemm.log
