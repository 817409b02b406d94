#!/bin/bash
# HBM traffic of every kernel at the shapes bench.py reports: the driver's --steps 20 launch group, the default 32 x 32-tile
# group + the one-batch call + the 4096-tile clip, and the four legs.  FETCH_SIZE and WRITE_SIZE need their own rocprofv3
# runs (TCC counter slots); --kernel-trace only, never combined with other trace domains.  -> gpurun_out/traffic.json
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-cli --no-parity-check --min-time 0.02 --max-rounds 6"
cd /tmp
run() { tag=$1; shift
  for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
    n=${c%%:*}; ctr=${c##*:}
    rm -rf $OUT/pmc_${n}_$tag
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${n}_$tag -o p -- "$@" > $OUT/pmc_${n}_$tag.json 2> $OUT/pmc_${n}_$tag.err
    echo "pmc $n $tag exit $?"
  done; }
# DCS_TRAFFIC_LEGS="leg ..." restricts the visit to those legs (merge the result with scripts/traffic_merge.py)
if [ -z "${DCS_TRAFFIC_LEGS:-}" ]; then
run k20 $B --steps 20 --warmup 5 --legs= --sat-tiles 0
run g32 $B --steps 32 --warmup 8 --streams 1 --legs= --sat-tiles 4096
fi
for leg in ${DCS_TRAFFIC_LEGS:-ikala bach10_f16 bach10_f32 score_informed}; do
  run leg_$leg $B --only-legs --legs $leg
done
cd $GRAFT_REPO_ROOT
python scripts/traffic_summary.py $OUT | tee $OUT/traffic_summary.txt | head -70
find $OUT -name "*.db" -delete; find $OUT -path "*pmc_*" -name "*kernel_trace.csv" -delete
