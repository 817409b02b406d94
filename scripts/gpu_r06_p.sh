#!/bin/bash
# Round 6, visit P: final_bf16x3_kernel with the co-resident workgroups' tile loops staggered (experiment build _exp_stagger.so)
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
: > $OUT/r06_p_stagger.txt
for rep in 1 2; do
for st in 0 2 4 7 10 14; do
  DCS_LIB=deepconvsep_amd/_exp_stagger.so DCS_FINAL_STAGGER=$st timeout 600 python bench.py --steps 20 --warmup 5 --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_p.line 2> $OUT/r06_p.err || tail -n 5 $OUT/r06_p.err
  python - "$st" <<'PY' | tee -a $OUT/r06_p_stagger.txt
import json, sys
d = json.load(open("bench_detail.json"))
k = d["launch_group"]["kernels_ms"]
print("stagger %2s x 256 clk: ms_per_step %.5f  frac %.4f  final %.1f us (rocprof-like avg %.1f)  parity %s | %s" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"], 1e3 * k["final"], 1e3 * d["roofline"]["avg_kernel_ms"], (d.get("parity_check") or {}).get("ok"), " ".join("%s %.1f" % (a, 1e3 * b) for a, b in k.items())))
PY
done
done
