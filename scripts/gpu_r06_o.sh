#!/bin/bash
# Round 6, visit O: the driver's --steps 20 round as ONE launch group against two / three / four groups on separate streams
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/r06_o_split.txt
for rep in 1 2; do
for cfg in "32 3" "10 2" "10 3" "7 3" "5 4" "14 2"; do
  set -- $cfg
  timeout 600 python bench.py --steps 20 --warmup 5 --clips-per-launch $1 --streams $2 --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 --no-parity-check > $OUT/r06_o.line 2> $OUT/r06_o.err || tail -n 5 $OUT/r06_o.err
  python - "$1" "$2" <<'PY' | tee -a $OUT/r06_o_split.txt
import json, sys
d = json.load(open("bench_detail.json"))
print("clips-per-launch %s streams %s: groups %s  ms_per_step %.5f  frac %.4f" % (sys.argv[1], sys.argv[2], d["config"]["launch_groups_per_round"], d["ms_per_step"], d["whole_path_frac_of_f32_peak"]))
PY
done
done
