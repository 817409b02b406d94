#!/bin/bash
# The whole GPU suite without -x (every failure listed), log kept under gpurun_out/
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
t0=$SECONDS
timeout 1500 python -m pytest tests -m gpu -q --timeout=600 > $OUT/pytest_gpu_all.log 2>&1
echo "exit $? after $((SECONDS - t0)) s"; tail -n 12 $OUT/pytest_gpu_all.log | cut -c1-220
