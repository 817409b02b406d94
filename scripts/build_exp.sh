#!/bin/bash
# Experiment build: deepconvsep_amd/_exp_<name>.so = libdcs with ONE translation unit recompiled with extra flags.
#   scripts/build_exp.sh <name> <file.hip> [-DFLAG ...]      (run deepconvsep_amd/csrc/build.sh first)
# Used through DCS_LIB=<path> (deepconvsep_amd/_lib.py); the .so files are git-ignored and travel with gpurun.
set -euo pipefail
name=$1; src=$2; shift 2
cd "$(dirname "$(readlink -f "$0")")/../deepconvsep_amd/csrc"
mkdir -p build/exp
obj=build/exp/${name}_${src%.hip}.o
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden "$@" -c "$src" -o "$obj"
objs=()
for s in api dsd_lat fft fft_wave tiling gemm gemm_bf16x3 gemm_f16 colconv_wreg colconv_x3 colconv_fwd_x3 slabconv_ps conv1_mfma deconv1_mfma dsd dsd_bf16x3 generic net score gather wavio; do
  if [ "$s.hip" = "$src" ]; then objs+=("$obj"); else objs+=("build/$s.o"); fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -Wl,--version-script=libdcs.map "${objs[@]}" -ldl -lpthread -o ../_exp_${name}.so
echo "built deepconvsep_amd/_exp_${name}.so ($*)"
