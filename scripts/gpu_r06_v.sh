#!/bin/bash
# Round 6, visit V: conv2 + BiasLayer + bottleneck DenseLayer folded into one affine map (B2fc), DCS_FOLD_CONV2 on / off
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 -p no:cacheprovider -k "dsd or separate or batch or ragged or clips or fused or stereo or random or guard or operator" > $OUT/r06_v_pytest.log 2>&1
echo "pytest exit $?"; tail -n 4 $OUT/r06_v_pytest.log | cut -c1-200
: > $OUT/r06_v_fold.txt
for rep in 1 2 3; do
for v in 1 0; do
  DCS_FOLD_CONV2=$v timeout 600 python bench.py --steps 20 --warmup 5 --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 4096 > $OUT/r06_v.line 2> $OUT/r06_v.err || tail -n 5 $OUT/r06_v.err
  python - "$v" <<'PY' | tee -a $OUT/r06_v_fold.txt
import json, sys
d = json.load(open("bench_detail.json"))
k = d["launch_group"]["kernels_ms"]
pc = d.get("parity_check") or {}
sat = d.get("saturating") or {}
print("DCS_FOLD_CONV2=%s: ms_per_step %.5f  frac %.4f  parity %s net %.3g outside %s | %s | sat %s" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"], pc.get("ok"), pc.get("network_output_max_err") or -1, pc.get("masked_bins_outside_1e4"), " ".join("%s %.1f" % (a, 1e3 * b) for a, b in k.items()), {a: round(b, 4) for a, b in (sat.get("kernels_ms") or {}).items() if a in ("conv2", "fc")}))
PY
done
done
for v in 1 0; do
  DCS_FOLD_CONV2=$v timeout 600 python bench.py --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_v.line 2> $OUT/r06_v.err || tail -n 5 $OUT/r06_v.err
  python - "$v" <<'PY' | tee -a $OUT/r06_v_fold.txt
import json, sys
d = json.load(open("bench_detail.json"))
k = d["launch_group"]["kernels_ms"]
print("default shape DCS_FOLD_CONV2=%s: ms_per_step %.5f  frac %.4f | %s" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"], " ".join("%s %.1f" % (a, 1e3 * b) for a, b in k.items())))
PY
done
