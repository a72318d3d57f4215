#!/bin/bash
# Build the C-ABI shared library for gfx950 (cross-compiles without a GPU).
set -e
cd "$(dirname "$0")"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-result"
mkdir -p build
pids=()
for f in gemm attention convpos rowops audio engine; do
  if [ ! -f build/$f.o ] || [ $f.hip -nt build/$f.o ] || [ -n "$(find . -maxdepth 1 -name '*.hpp' -newer build/$f.o)" ] || [ ../../include/f5tts_hip.h -nt build/$f.o ]; then
    $HIPCC $FLAGS -c $f.hip -o build/$f.o &
    pids+=($!)
  fi
done
for p in "${pids[@]}"; do wait $p; done
$HIPCC --offload-arch=gfx950 -shared -fPIC build/gemm.o build/attention.o build/convpos.o build/rowops.o build/audio.o build/engine.o -o libf5tts_hip.so
echo "built $(pwd)/libf5tts_hip.so"
