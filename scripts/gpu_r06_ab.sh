#!/bin/bash
# Round 6, visit AB: the --steps 20 round replayed as a hipGraph (default) against eager launches (DCS_GRAPH=0)
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/r06_ab.txt
for rep in 1 2 3; do
for v in 1 0; do
  DCS_GRAPH=$v timeout 600 python bench.py --steps 20 --warmup 5 --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 --no-parity-check > $OUT/r06_ab.line 2> $OUT/r06_ab.err || tail -n 5 $OUT/r06_ab.err
  python - "$v" <<'PY' | tee -a $OUT/r06_ab.txt
import json, sys
d = json.load(open("bench_detail.json"))
k = d["launch_group"]["kernels_ms"]
print("DCS_GRAPH=%s: ms_per_step %.5f  frac %.4f  round %.1f us  kernels (events) sum %.1f us" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"], 20e3 * d["ms_per_step"], 1e3 * sum(k.values())))
PY
done
done
