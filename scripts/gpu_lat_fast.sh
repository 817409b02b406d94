#!/bin/bash
# Short GPU-box visit while iterating on the one-batch kernels: their tests, us/step of both families, in-kernel timeline.
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/mask_bins.txt
timeout 900 python -m pytest tests/test_gpu_latency.py -m gpu -q -x --timeout=300 -p no:cacheprovider 2>&1 | tail -n 15
DCS_LAT_EXP_STAGES=${DCS_LAT_EXP_STAGES:-0,255,511} timeout 600 python scripts/gpu_lat_exp.py 2>&1 | grep -v amdgpu.ids | tee $OUT/lat_exp_fast.log
DCS_LAT_EXP_N=1024 DCS_LAT_EXP_STAGES=0,255,511 timeout 600 python scripts/gpu_lat_exp.py 2>&1 | grep -E "^stages" | sed 's/^/N=1024 /'
if [ -f deepconvsep_amd/_exp_lattrace.so ]; then
  DCS_LIB=$PWD/deepconvsep_amd/_exp_lattrace.so timeout 300 python scripts/gpu_lat_trace.py 2>&1 | grep -v amdgpu.ids | tee $OUT/lat_trace.txt
fi
