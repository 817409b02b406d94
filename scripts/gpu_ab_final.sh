#!/bin/bash
# A/B of libdcs builds on the headline shapes: final / deconv2 kernel times of a 32 x 32-tile launch group and of the
# 4096-tile clip (HIP events), the headline value, the parity check.   DCS_AB_LIBS="default _exp_x.so ..."
export TMPDIR=/tmp
mkdir -p gpurun_out
for lib in ${DCS_AB_LIBS:-default}; do
  if [ "$lib" = "default" ]; then unset DCS_LIB; else export DCS_LIB=$PWD/deepconvsep_amd/$lib; fi
  for rep in 1 2; do
  timeout 300 python bench.py --no-cpu-baseline --no-host-fed --legs= --no-cli --min-time 0.15 > gpurun_out/ab_$lib.json 2> gpurun_out/ab_$lib.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/ab_$lib.json").read().strip().splitlines()[-1])
g=d["launch_group"]["kernels_ms"]; s=d["saturating"]["kernels_ms"]
f=lambda k: " ".join("%s %.1f" % (t, 1e3*k[t]) for t in ("stft","deconv2","final","istft"))
print("%-18s value %.2fM parity %s | group(us): %s | sat(us): %s | sat ms %.4f | roofline frac %.3f (%.1f us)" % ("$lib", d["value"]/1e6, d["parity_check"]["ok"], f(g), f(s), d["saturating"]["ms_per_step"], d["roofline"]["frac"], 1e3*d["roofline"]["avg_kernel_ms"]))
PY
  done
done
