#!/bin/bash
# Round 6, visit B: the encoder GEMMs of the DSD path on gemm_ks_kernel (K split over the waves of a workgroup, bf16 x 3):
# DSD parity tests with the kernel on, then the driver's command per configuration (DCS_GEMM_KS=0 off | 1 | 2 | 3), alternating,
# with the per-kernel HIP-event times of the launch group, the saturating clip and the default shape.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rm -f $OUT/mask_bins.txt
echo "== DSD parity tests (kernel on)"
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 --timeout=400 -p no:cacheprovider -k "${DCS_B_K:-dsd or fused or batch or whole_path or bf16x3 or latency or random or guard or stereo or ragged or smoke}" > $OUT/r06_b_pytest.log 2>&1
echo "pytest exit $?"; tail -n 12 $OUT/r06_b_pytest.log | cut -c1-220
: > $OUT/r06_b_gemm_ks_ab.txt
for v in ${DCS_B_VARIANTS:-0 1 2 3 0 1 2 3}; do
  DCS_GEMM_KS=$v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs "" --sat-tiles 4096 --no-host-fed --no-cli > $OUT/r06_b.line 2> $OUT/r06_b.err || tail -n 5 $OUT/r06_b.err
  python - "$v" <<'PY' | tee -a $OUT/r06_b_gemm_ks_ab.txt
import json, sys
d = json.load(open("bench_detail.json"))
g = d["launch_group"]; s = d.get("saturating") or {}
print("KS=%s k20: %.5f ms/step whole %.4f | group %s sum %.4f | pcm err %.2e ok %s" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"],
      " ".join("%s %.4f" % kv for kv in g["kernels_ms"].items()), g["kernels_ms_sum"], d["parity_check"]["max_abs_pcm_err"], d["parity_check"]["ok"]))
if s: print("      sat 4096: %.4f ms | %s" % (s["ms_per_step"], " ".join("%s %.4f" % kv for kv in s["kernels_ms"].items())))
PY
done
for v in ${DCS_B_DEFAULT_VARIANTS:-0 1 0 1}; do
  DCS_GEMM_KS=$v timeout 600 python bench.py --no-cpu-baseline --legs "" --sat-tiles 0 --no-host-fed --no-cli > $OUT/r06_b.line 2> $OUT/r06_b.err || tail -n 5 $OUT/r06_b.err
  python - "$v" <<'PY' | tee -a $OUT/r06_b_gemm_ks_ab.txt
import json, sys
d = json.load(open("bench_detail.json"))
g = d["launch_group"]
print("KS=%s default shape: %.5f ms/step whole %.4f | group %s sum %.4f" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"],
      " ".join("%s %.4f" % kv for kv in g["kernels_ms"].items()), g["kernels_ms_sum"]))
PY
done
