#!/bin/bash
# Round 5: HBM traffic of the legs whose kernels changed (DCS_TRAFFIC_LEGS, default score_informed; two --pmc passes per leg,
# --kernel-trace only).
#   -> gpurun_out/traffic.json; then python scripts/traffic_merge.py gpurun_out/traffic.json profiles/r05_traffic.json
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
rm -rf $OUT/pmc_*
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-cli --no-parity-check --min-time 0.02 --max-rounds 6"
cd /tmp
for leg in ${DCS_TRAFFIC_LEGS:-score_informed}; do
for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
  n=${c%%:*}; ctr=${c##*:}
  timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${n}_leg_$leg -o p -- $B --only-legs --legs $leg \
      > $OUT/pmc_${n}_leg_$leg.json 2> $OUT/pmc_${n}_leg_$leg.err
  echo "pmc $n $leg exit $?"
done
done
cd $GRAFT_REPO_ROOT
python scripts/traffic_summary.py $OUT | tee $OUT/traffic_summary_si.txt | tail -60
find $OUT -name "*.db" -delete; find $OUT -path "*pmc_*" -name "*kernel_trace.csv" -delete
