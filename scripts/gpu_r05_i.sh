#!/bin/bash
# Round 5, visit I: iSTFT chain kernel with the next frame's rows requested before the FFT (DCS_ISTFT_CHAIN_PIPE=0: without).
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider \
  -k "istft or inverse or roundtrip or bench_launch_shapes or separate_batch_equals or separate_ragged or kernel_variants or chain or pcm16 or batch_driver" > $OUT/r05_i_pytest.log 2>&1
echo "pytest exit $?"; tail -n 6 $OUT/r05_i_pytest.log | cut -c1-200
DCS_AB_VARIANTS="default DCS_ISTFT_CHAIN_PIPE=0 default DCS_ISTFT_CHAIN_PIPE=0" DCS_K20_REPS=1 DCS_K20_TRACE=1 bash scripts/gpu_k20_ab.sh
for v in default DCS_ISTFT_CHAIN_PIPE=0; do
  envs=""; [ "$v" != "default" ] && envs="$v"
  env $envs timeout 600 python bench.py --legs= --no-cpu-baseline --no-host-fed --no-cli --no-parity-check --sat-tiles 4096 > $OUT/r05_i_$v.line 2> $OUT/r05_i_$v.err
  python - "$v" <<'PY'
import json, sys
d = json.load(open("bench_detail.json"))
print("%-28s default shape %.5f ms/step whole %.3f | group istft %.1f us | sat %.4f ms istft %.1f us | single %.4f" % (
    sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"], 1e3 * d["launch_group"]["kernels_ms"]["istft"],
    d["saturating"]["ms_per_step"], 1e3 * d["saturating"]["kernels_ms"]["istft"], d["single_stream_ms_per_step"]))
PY
done
