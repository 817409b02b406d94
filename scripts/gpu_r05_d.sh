#!/bin/bash
# Round 5, visit D: the whole GPU suite after the prune, smoke, the driver's command, the k20 kernel trace.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/mask_bins.txt
t0=$(date +%s)
timeout 2400 python -m pytest tests -m gpu -q -x --timeout=900 -p no:cacheprovider --durations=15 > $OUT/r05_d_pytest.log 2>&1
echo "pytest exit $? in $(( $(date +%s) - t0 )) s"; tail -n 25 $OUT/r05_d_pytest.log
cp $OUT/mask_bins.txt $OUT/r05_d_mask_bins.txt 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_d_driver_stdout.txt 2> $OUT/r05_d_driver.err; echo "driver cmd exit $?"
cp bench_detail.json $OUT/r05_d_bench_detail.json
tail -c 2600 $OUT/r05_d_driver_stdout.txt
DCS_AB_VARIANTS="default" DCS_K20_REPS=1 DCS_K20_TRACE=1 bash scripts/gpu_k20_ab.sh
