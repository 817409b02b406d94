#!/bin/bash
# Round 5: HBM traffic (FETCH_SIZE / WRITE_SIZE, separate rocprofv3 --pmc passes, --kernel-trace only) of the DSD launch shapes
# bench.py reports -- the driver's --steps 20 group and the default 32 x 32 group + one batch + the 4096-tile clip.  The legs'
# kernels did not change in round 5: their records are carried over from profiles/r04_traffic.json.
#   -> gpurun_out/traffic.json; then: cp profiles/r04_traffic.json profiles/r05_traffic.json;
#      python scripts/traffic_merge.py gpurun_out/traffic.json profiles/r05_traffic.json
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
run k20 $B --steps 20 --warmup 5 --legs= --sat-tiles 0
run g32 $B --steps 32 --warmup 8 --streams 1 --legs= --sat-tiles 4096
cd $GRAFT_REPO_ROOT
python scripts/traffic_summary.py $OUT | tee $OUT/traffic_summary.txt | head -60
find $OUT -name "*.db" -delete; find $OUT -path "*pmc_*" -name "*kernel_trace.csv" -delete
