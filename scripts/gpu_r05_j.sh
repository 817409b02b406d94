#!/bin/bash
# Round 5, visit J: batch driver with the device work enqueued and collected one iteration later (main thread never waits
# for the device): byte identity, then the steady-state figure over a few settings.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -k "batch_driver or pcm16 or separate_many" > $OUT/r05_j_pytest.log 2>&1
echo "pytest exit $?"; tail -n 4 $OUT/r05_j_pytest.log | cut -c1-200
timeout 900 python scripts/batch_driver_bench.py --reps 2 --variants "-w 16 -g 16" "-w 16 -g 32" "-w 32 -g 16" "-w 8 -g 16" "-w 16 -g 8" 2>&1 | tee $OUT/r05_j_batch_driver.txt | cut -c1-250
nproc; df -h /tmp | tail -1
