#!/bin/bash
# The GPU suite as the driver runs it, with the 30 slowest tests listed
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
t0=$SECONDS
timeout 1500 python -m pytest tests/ -x -q -m gpu --durations=30 > $OUT/pytest_gpu_durations.log 2>&1
echo "exit $? after $((SECONDS - t0)) s"; grep -A34 "slowest" $OUT/pytest_gpu_durations.log | cut -c1-160; tail -n 2 $OUT/pytest_gpu_durations.log
