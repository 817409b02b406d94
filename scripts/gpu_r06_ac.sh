#!/bin/bash
# Round 6, visit AC: graph replay against eager launches at the default shape (12 launch groups of 32 steps over 3 streams) and at 64 / 8 steps
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/r06_ac.txt
for args in "" "--steps 64 --warmup 8" "--steps 8 --warmup 4" "--steps 4 --warmup 4" "--steps 2 --warmup 4"; do
for v in 1 0; do
  DCS_GRAPH=$v timeout 600 python bench.py $args --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 --no-parity-check > $OUT/r06_ac.line 2> $OUT/r06_ac.err || tail -n 5 $OUT/r06_ac.err
  python - "$v" "$args" <<'PY' | tee -a $OUT/r06_ac.txt
import json, sys
d = json.load(open("bench_detail.json"))
print("DCS_GRAPH=%s %-24s groups %s: ms_per_step %.5f  frac %.4f" % (sys.argv[1], sys.argv[2] or "(default)", d["config"]["launch_groups_per_round"], d["ms_per_step"], d["whole_path_frac_of_f32_peak"]))
PY
done
done
