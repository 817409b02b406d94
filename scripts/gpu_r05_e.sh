#!/bin/bash
# Round 5, visit E: the whole GPU suite (no -x: every failure at once), then A/B of the pruned final kernel against the pre-prune
# object on the same box (alternating), then the k20 trace.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/mask_bins.txt
t0=$(date +%s)
timeout 2700 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider --durations=12 > $OUT/r05_e_pytest.log 2>&1
echo "pytest exit $? in $(( $(date +%s) - t0 )) s"; tail -n 30 $OUT/r05_e_pytest.log | cut -c1-220
cp $OUT/mask_bins.txt $OUT/r05_e_mask_bins.txt 2>/dev/null
DCS_AB_VARIANTS="default DCS_LIB=deepconvsep_amd/_exp_oldfinal.so default DCS_LIB=deepconvsep_amd/_exp_oldfinal.so" DCS_K20_REPS=1 DCS_K20_TRACE=1 bash scripts/gpu_k20_ab.sh
