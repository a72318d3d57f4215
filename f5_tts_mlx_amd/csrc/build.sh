#!/bin/bash
# Build the C-ABI shared library for gfx950 (cross-compiles without a GPU).
# The kernel sources are compiled twice: bf16 MFMA operands (namespace f5bf) and fp16 operands (-DF5_F16=1, namespace f5hf).
#   bash build.sh            product library: the kernels sample() / the vocoder / the mel front-end can reach
#   F5_LAB=1 bash build.sh   lab library (same file name): + the superseded / rejected kernels and the hooks that select them
#                            (include/f5tts_hip_lab.h).  Objects of the two flavours live in build/ and build_lab/.
#   F5_PROBE=1 bash build.sh measurement build libf5tts_hip_probe.so (objects in build_probe/): the product kernels + the ablation
#                            switches of csrc/gemm_dev.hpp F5_PROBE_* (epilogue without stores / without its arithmetic, phase-shifted
#                            first round, non-temporal residual stream); loaded through F5TTS_HIP_LIB by tools/r5_epilogue_probe.py only
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
LAB=${F5_LAB:-0}
PROBE=${F5_PROBE:-0}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result -DF5_LAB=$LAB -DF5_PROBE=$PROBE"
B=build
OUT=libf5tts_hip.so
if [ "$PROBE" = 1 ]; then B=build_probe; OUT=libf5tts_hip_probe.so; fi
KERNELS="gemm gemm256 gemm_rs128 gemm_f8 attention convpos rowops"
if [ "$LAB" = 1 ]; then B=build_lab; KERNELS="$KERNELS gemm_lab gemm128"; fi
mkdir -p $B
pids=()
stale() {  # $1 = source, $2 = object, $3 = 1 when the source includes the public C-ABI header
  [ ! -f "$2" ] || [ "$1" -nt "$2" ] || [ -n "$(find . -maxdepth 1 -name '*.hpp' -newer "$2")" ] || { [ "$3" = 1 ] && [ ../../include/f5tts_hip.h -nt "$2" ]; }
}
# The GEMM files are built WITHOUT the SLP vectoriser.  On gfx950 a packed-f32 VALU instruction whose LO result reads the HI
# register of src1 (v_pk_mul_f32 / v_pk_add_f32 ... op_sel:[x,1]) returns that operand as 0 on lanes 48-63 now and then when
# another wave of the same SIMD has MFMAs in flight (tools/probes/pk_f32_vs_mfma2.hip, DESIGN.md "packed f32 next to MFMA").  The
# SLP vectoriser produces exactly that form from the rotation / gate arithmetic of the GEMM epilogues, which run next to other
# workgroups' (or the other wave group's) MFMA loops.  tests/test_isa.py checks the ISA of every kernel that contains MFMAs.
# rowops.hip follows because it shares lnrow.hpp with the LN tail fused into gemm.hip: the two must produce the same bits.
noslp() { case "$1" in gemm|gemm256|gemm_rs128|gemm_f8|gemm_lab|gemm128|rowops) echo "-fno-slp-vectorize";; *) echo "";; esac; }
objs=()
for f in $KERNELS; do
  for v in 0 1; do
    o=$B/${f}_h$v.o
    objs+=($o)
    if stale $f.hip $o 0 || [ build.sh -nt $o ]; then $HIPCC $FLAGS $(noslp $f) -DF5_F16=$v -c $f.hip -o $o & pids+=($!); fi
  done
done
for f in audio vocoder noise engine; do
  o=$B/$f.o
  objs+=($o)
  if stale $f.hip $o 1; then $HIPCC $FLAGS -c $f.hip -o $o & pids+=($!); fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC "${objs[@]}" -o $OUT
echo "built $(pwd)/$OUT (F5_LAB=$LAB F5_PROBE=$PROBE)"
