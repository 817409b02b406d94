#!/bin/bash
# Round 6, visit Q: final_bf16x3_kernel walking several row groups per workgroup (one resident round), DCS_FINAL_UPW sweep
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -x -q --timeout=600 -p no:cacheprovider -k "dsd or separate or batch or ragged or clips or fused or stereo" > $OUT/r06_q_pytest.log 2>&1
echo "pytest exit $?"; tail -n 4 $OUT/r06_q_pytest.log | cut -c1-200
: > $OUT/r06_q_upw.txt
for rep in 1 2; do
for u in 1 2 3 0 4; do
  DCS_FINAL_UPW=$u timeout 600 python bench.py --steps 20 --warmup 5 --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 4096 > $OUT/r06_q.line 2> $OUT/r06_q.err || tail -n 5 $OUT/r06_q.err
  python - "$u" <<'PY' | tee -a $OUT/r06_q_upw.txt
import json, sys
d = json.load(open("bench_detail.json"))
k = d["launch_group"]["kernels_ms"]
sat = d.get("saturating") or {}
print("row groups per workgroup %s (0 = auto): ms_per_step %.5f  frac %.4f  final %.1f us (avg %.1f)  parity %s | sat clip %s ms final %s" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"], 1e3 * k["final"], 1e3 * d["roofline"]["avg_kernel_ms"], (d.get("parity_check") or {}).get("ok"), sat.get("ms_per_clip"), (sat.get("kernels_ms") or {}).get("final")))
PY
done
done
for u in 1 0; do
  DCS_FINAL_UPW=$u timeout 600 python bench.py --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_q.line 2> $OUT/r06_q.err || tail -n 5 $OUT/r06_q.err
  python - "$u" <<'PY' | tee -a $OUT/r06_q_upw.txt
import json, sys
d = json.load(open("bench_detail.json"))
k = d["launch_group"]["kernels_ms"]
print("default shape, row groups per workgroup %s: ms_per_step %.5f  frac %.4f  final %.1f us" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"], 1e3 * k["final"]))
PY
done
