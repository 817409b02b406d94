#!/bin/bash
# Round 6, visit AD: the folded bottleneck GEMM with its weights in the few-rows kernel's fragment order (one 16-byte B load per lane and 16 K)
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 -p no:cacheprovider -k "dsd or separate or batch or ragged or clips or fused or stereo or variants or guard or launch_shapes or adversarial or random" > $OUT/r06_ad_pytest.log 2>&1
echo "pytest exit $?"; tail -n 3 $OUT/r06_ad_pytest.log | cut -c1-200
: > $OUT/r06_ad.txt
for rep in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_ad.line 2> $OUT/r06_ad.err || tail -n 5 $OUT/r06_ad.err
  python - <<'PY' | tee -a $OUT/r06_ad.txt
import json
d = json.load(open("bench_detail.json"))
k = d["launch_group"]["kernels_ms"]
pc = d.get("parity_check") or {}
print("steps 20: ms_per_step %.5f  frac %.4f  parity %s net %.3g | %s" % (d["ms_per_step"], d["whole_path_frac_of_f32_peak"], pc.get("ok"), pc.get("network_output_max_err") or -1, " ".join("%s %.1f" % (a, 1e3 * b) for a, b in k.items())))
PY
done
timeout 600 python bench.py --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_ad.line 2> $OUT/r06_ad.err || tail -n 5 $OUT/r06_ad.err
python - <<'PY' | tee -a $OUT/r06_ad.txt
import json
d = json.load(open("bench_detail.json"))
k = d["launch_group"]["kernels_ms"]
print("default shape: ms_per_step %.5f  frac %.4f | %s" % (d["ms_per_step"], d["whole_path_frac_of_f32_peak"], " ".join("%s %.1f" % (a, 1e3 * b) for a, b in k.items())))
PY
