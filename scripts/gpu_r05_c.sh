#!/bin/bash
# Round 5, visit C: chained iSTFT (parity + A/B at the driver's shape), int16 batch driver, batched compute_transform.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 -p no:cacheprovider \
  -k "${DCS_C_K:-staged_istft or bench_launch_shapes or separate_batch_equals or compute_transform or pcm16 or batch_driver or pcm_to_int16 or separate_ragged or (kernel_variants and (env20 or env21 or env22)) or stft or istft or roundtrip}" > $OUT/r05_c_pytest.log 2>&1
echo "pytest exit $?"; tail -n 15 $OUT/r05_c_pytest.log
DCS_AB_VARIANTS="default DCS_ISTFT_CHAIN=0 DCS_ISTFT_CHAIN=7 DCS_ISTFT_CHAIN=9" DCS_K20_REPS=1 DCS_K20_TRACE=0 bash scripts/gpu_k20_ab.sh
# the default shape (32 x 32 tiles x 3 streams) and the transform / cli legs
timeout 900 python bench.py --legs transform --no-cpu-baseline --no-host-fed --sat-tiles 4096 > $OUT/r05_c_bench.line 2> $OUT/r05_c_bench.err; echo "bench exit $?"; tail -n 3 $OUT/r05_c_bench.err
cp bench_detail.json $OUT/r05_c_bench_detail.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_c_bench_detail.json"))
print("default shape: %.5f ms/step, %.2f M frames/s, whole %.3f" % (d["ms_per_step"], d["value"] / 1e6, d["whole_path_frac_of_f32_peak"]))
print("group us:", {k: round(1e3 * v, 1) for k, v in d["launch_group"]["kernels_ms"].items()})
print("sat:", d["saturating"]["ms_per_step"], {k: round(1e3 * v, 1) for k, v in d["saturating"]["kernels_ms"].items()})
print(json.dumps(d["legs"]["transform"], indent=1))
print(json.dumps(d["cli"], indent=1))
PY
