#!/bin/bash
# Round 5: HBM traffic of the score-informed leg after its decoder moved to the fused kernel (two --pmc passes, --kernel-trace only).
#   -> gpurun_out/traffic.json; then python scripts/traffic_merge.py gpurun_out/traffic.json profiles/r05_traffic.json
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
rm -rf $OUT/pmc_*
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-cli --no-parity-check --min-time 0.02 --max-rounds 6"
cd /tmp
for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
  n=${c%%:*}; ctr=${c##*:}
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${n}_leg_score_informed -o p -- $B --only-legs --legs score_informed \
      > $OUT/pmc_${n}_leg_score_informed.json 2> $OUT/pmc_${n}_leg_score_informed.err
  echo "pmc $n exit $?"
done
cd $GRAFT_REPO_ROOT
python scripts/traffic_summary.py $OUT | tee $OUT/traffic_summary_si.txt | tail -25
find $OUT -name "*.db" -delete; find $OUT -path "*pmc_*" -name "*kernel_trace.csv" -delete
