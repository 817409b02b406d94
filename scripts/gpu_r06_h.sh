#!/bin/bash
# Round 6, visit H: forward STFT with twelve frames per workgroup and the window in LDS -- full GPU suite (also covers the
# deleted switches / the pruned decoder branch), the driver's command per variant (default vs DCS_STFT_FPW=4, alternating),
# the in-kernel timeline at the driver's shape.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rm -f $OUT/mask_bins.txt $OUT/f16_stats.txt
timeout 1500 python -m pytest tests -m gpu -q --maxfail=30 --timeout=600 -p no:cacheprovider --durations=12 > $OUT/r06_h_pytest.log 2>&1
echo "pytest exit $?"; tail -n 22 $OUT/r06_h_pytest.log | cut -c1-220
: > $OUT/r06_h_stft_ab.txt
for v in default DCS_STFT_FPW=4 default DCS_STFT_FPW=4 default DCS_STFT_FPW=4; do
  envs=""; [ "$v" != "default" ] && envs="$v"
  env $envs timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs "" --sat-tiles 4096 --no-host-fed --no-cli > $OUT/r06_h.line 2> $OUT/r06_h.err || tail -n 5 $OUT/r06_h.err
  python - "$v" <<'PY' | tee -a $OUT/r06_h_stft_ab.txt
import json, sys
d = json.load(open("bench_detail.json"))
g = d["launch_group"]; s = d.get("saturating") or {}
print("%-16s k20: %.5f ms/step whole %.4f | group %s sum %.4f | hbm %s | pcm err %.2e ok %s" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"],
      " ".join("%s %.4f" % kv for kv in g["kernels_ms"].items()), g["kernels_ms_sum"], {k: v["frac"] for k, v in (d.get("hbm_stages") or {}).items()}, d["parity_check"]["max_abs_pcm_err"], d["parity_check"]["ok"]))
if s: print("      sat 4096: %.4f ms | %s" % (s["ms_per_step"], " ".join("%s %.4f" % kv for kv in s["kernels_ms"].items())))
PY
done
for v in default DCS_STFT_FPW=4 default DCS_STFT_FPW=4; do
  envs=""; [ "$v" != "default" ] && envs="$v"
  env $envs timeout 600 python bench.py --no-cpu-baseline --legs "" --sat-tiles 0 --no-host-fed --no-cli > $OUT/r06_h.line 2> $OUT/r06_h.err || tail -n 5 $OUT/r06_h.err
  python - "$v" <<'PY' | tee -a $OUT/r06_h_stft_ab.txt
import json, sys
d = json.load(open("bench_detail.json"))
g = d["launch_group"]
print("%-16s default shape: %.5f ms/step whole %.4f | group %s sum %.4f" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"],
      " ".join("%s %.4f" % kv for kv in g["kernels_ms"].items()), g["kernels_ms_sum"]))
PY
done
DCS_TRACE_CLIPS=20 DCS_TRACE_TILES=32 DCS_LIB=deepconvsep_amd/_exp_fftwtrace.so timeout 300 python scripts/gpu_fftw_trace.py 2>&1 | grep -v amdgpu.ids | tee $OUT/r06_h_stft_timeline.txt
