#!/bin/bash
# Round 5, visit U: rocprofv3 kernel trace of the generic-graph legs on the final build, summarised by kernel and grid.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd /tmp
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_legs -o legs -- \
  python $GRAFT_REPO_ROOT/bench.py --only-legs --legs score_informed,bach10_f32,bach10_f16,ikala --no-cpu-baseline --no-host-fed --no-cli --no-parity-check \
  > $OUT/prof_legs.line 2> $OUT/prof_legs.err
echo "rocprof exit $?"
cd $GRAFT_REPO_ROOT
python scripts/trace_by_grid.py $OUT/prof_legs > $OUT/kernel_durations_by_grid_legs.txt 2>&1
grep -E "conv1_mfma|skinny|longk|fused|slabconv|split_a|colconv" $OUT/kernel_durations_by_grid_legs.txt | head -30
rm -rf $OUT/prof_legs
