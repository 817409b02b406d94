#!/bin/bash
# Round 6, visit D: conv1 / conv2 of the DSD launch group with K split over workgroups and waves (gemm_ks.hip, second form):
# DSD parity tests with it on, the driver's command per variant (alternating), in-kernel timelines of the two layers.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rm -f $OUT/mask_bins.txt
echo "== DSD parity tests (kernel on)"
timeout 1200 python -m pytest tests -m gpu -q --maxfail=20 --timeout=400 -p no:cacheprovider -k "${DCS_B_K:-dsd or fused or batch or whole_path or bf16x3 or latency or random or guard or stereo or ragged or smoke}" > $OUT/r06_d_pytest.log 2>&1
echo "pytest exit $?"; tail -n 12 $OUT/r06_d_pytest.log | cut -c1-220
: > $OUT/r06_d_gemm_ks_ab.txt
for v in ${DCS_D_VARIANTS:-DCS_GEMM_KS=0 DCS_GEMM_KS=1 DCS_GEMM_KS_RB=4 DCS_GEMM_KS_RB=2 DCS_GEMM_KS=0 DCS_GEMM_KS=1 DCS_GEMM_KS_RB=4 DCS_GEMM_KS_RB=2}; do
  env $v timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs "" --sat-tiles 4096 --no-host-fed --no-cli > $OUT/r06_d.line 2> $OUT/r06_d.err || tail -n 5 $OUT/r06_d.err
  python - "$v" <<'PY' | tee -a $OUT/r06_d_gemm_ks_ab.txt
import json, sys
d = json.load(open("bench_detail.json"))
g = d["launch_group"]; s = d.get("saturating") or {}
print("%-18s k20: %.5f ms/step whole %.4f | group %s sum %.4f | pcm err %.2e ok %s" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"],
      " ".join("%s %.4f" % kv for kv in g["kernels_ms"].items()), g["kernels_ms_sum"], d["parity_check"]["max_abs_pcm_err"], d["parity_check"]["ok"]))
if s: print("      sat 4096: %.4f ms | %s" % (s["ms_per_step"], " ".join("%s %.4f" % kv for kv in s["kernels_ms"].items())))
PY
done
for v in DCS_GEMM_KS=0 DCS_GEMM_KS=1 DCS_GEMM_KS=0 DCS_GEMM_KS=1; do
  env $v timeout 600 python bench.py --no-cpu-baseline --legs "" --sat-tiles 0 --no-host-fed --no-cli > $OUT/r06_d.line 2> $OUT/r06_d.err || tail -n 5 $OUT/r06_d.err
  python - "$v" <<'PY' | tee -a $OUT/r06_d_gemm_ks_ab.txt
import json, sys
d = json.load(open("bench_detail.json"))
g = d["launch_group"]
print("%-18s default shape: %.5f ms/step whole %.4f | group %s sum %.4f" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"],
      " ".join("%s %.4f" % kv for kv in g["kernels_ms"].items()), g["kernels_ms_sum"]))
PY
done
: > $OUT/r06_d_ks_timeline.txt
for kv in conv1=1028 conv2=780; do
  name=${kv%=*}
  DCS_KS_LAYER=$name DCS_LIB=deepconvsep_amd/_exp_kstrace_$name.so timeout 300 python scripts/gpu_ks_trace.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/r06_d_ks_timeline.txt
done
