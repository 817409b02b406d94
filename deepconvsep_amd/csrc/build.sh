#!/bin/bash
# Build libdcs.so for gfx950 (MI355X).  hipcc cross-compiles without a GPU.
set -euo pipefail
cd "$(dirname "$0")"
OUT=../libdcs.so
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -fvisibility=default \
    api.hip fft.hip fft_wave.hip tiling.hip gemm.hip dsd.hip dsd_bf16x3.hip generic.hip net.hip score.hip -o "$OUT"
echo "built $(readlink -f "$OUT")"
