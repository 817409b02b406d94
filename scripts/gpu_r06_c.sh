#!/bin/bash
# Round 6, visit C: where do the 12 us of a gemm_ks_kernel workgroup go?  In-kernel timelines of conv1 / conv2 / fc / fc1x at the
# driver's shape (experiment builds, one per layer).
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/r06_c_ks_timeline.txt
for kv in conv1=1028 conv2=780 fc=832 fc1x=128; do
  name=${kv%=*}
  DCS_KS_LAYER=$name DCS_LIB=deepconvsep_amd/_exp_kstrace_$name.so timeout 300 python scripts/gpu_ks_trace.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/r06_c_ks_timeline.txt
done
