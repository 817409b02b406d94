#!/bin/bash
# Round 6, visit S: which chained iSTFT runs (kernel trace) and for how long, lean on / off
set -u
cd /tmp 2>/dev/null; cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
for v in 1 0; do
  rm -rf $OUT/prof_s_$v
  DCS_ISTFT_LEAN=$v timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_s_$v -o s -- python bench.py --steps 20 --warmup 5 --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 --max-rounds 200 > $OUT/r06_s_$v.line 2> $OUT/r06_s_$v.err
  f=$(find $OUT/prof_s_$v -name "*kernel_stats.csv" | head -1)
  echo "== DCS_ISTFT_LEAN=$v  $f"; head -12 "$f" | cut -c1-220
  t=$(find $OUT/prof_s_$v -name "*kernel_trace.csv" | head -1)
  python scripts/trace_by_grid.py "$t" 2>/dev/null | grep -i "istft\|void" | head -5
  cp "$f" $OUT/r06_s_kernel_stats_lean$v.csv
  rm -rf $OUT/prof_s_$v
done
