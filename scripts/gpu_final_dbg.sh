#!/bin/bash
# Experiment: where does final_kernel's time go?  Variant libs built with -DDCS_FINAL_DBG=k (see dsd.hip)
# are prebuilt HERE (cross-compile) as deepconvsep_amd/libdcs_dbg<k>.so and timed on the 4096-tile leg.
set -u
OUT=gpurun_out; mkdir -p $OUT
for k in 0 1 2 3 4 5; do
  lib=$PWD/deepconvsep_amd/libdcs_dbg$k.so
  [ -f $lib ] || continue
  DCS_LIB=$lib timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/dbg_$k.json 2> $OUT/dbg_$k.err
  python - <<PY
import json
d=json.load(open("$OUT/dbg_$k.json")); s=d['saturating']
print("dbg $k: SAT ms/step %.4f final %.4f" % (s['ms_per_step'], s['kernels_ms']['final']), "32t single final %.4f" % d['single_stream']['kernels_ms']['final'])
PY
done
