#!/bin/bash
# Build libdcs.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
# Every translation unit is compiled to build/<name>.o (in parallel, only when it or a header changed), then linked.
set -euo pipefail
SELF=$(readlink -f "$0")
cd "$(dirname "$SELF")"
OUT=../libdcs.so
SRCS="api.hip dsd_lat.hip fft.hip fft_wave.hip tiling.hip gemm.hip gemm_bf16x3.hip gemm_f16.hip colconv_wreg.hip colconv_x3.hip colconv_fwd_x3.hip slabconv_ps.hip conv1_mfma.hip deconv1_mfma.hip dsd.hip dsd_bf16x3.hip generic.hip net.hip score.hip gather.hip wavio.hip"
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden"
mkdir -p build
newest_header=$(ls -t *.h ../../include/*.h | head -1)
pids=()
objs=()
for s in $SRCS; do
  o=build/${s%.hip}.o
  objs+=("$o")
  if [ ! -f "$o" ] || [ "$s" -nt "$o" ] || [ "$newest_header" -nt "$o" ] || [ "$SELF" -nt "$o" ]; then
    hipcc $FLAGS -c "$s" -o "$o" &
    pids+=($!)
  fi
done
for p in ${pids[@]+"${pids[@]}"}; do wait "$p"; done
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=libdcs.map "${objs[@]}" -ldl -lpthread -o "$OUT"
echo "built $(readlink -f "$OUT")"
