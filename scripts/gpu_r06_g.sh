#!/bin/bash
# Round 6, visit G: iKala conv2^T on column strips (slabconv_ps.hip) -- iKala tests, legs A/B (default vs DCS_SLABCONV_STRIP=0,
# alternating) -- and the in-kernel timeline of the forward STFT at the driver's launch shape (20 x 32 tiles).
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rm -f $OUT/mask_bins.txt
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout=600 -p no:cacheprovider -k "${DCS_G_K:-ikala or random or generic_graphs or guard or fallbacks}" > $OUT/r06_g_pytest.log 2>&1
echo "pytest exit $?"; tail -n 6 $OUT/r06_g_pytest.log | cut -c1-250
: > $OUT/r06_g_ikala_ab.txt
for v in default DCS_SLABCONV_STRIP=0 default DCS_SLABCONV_STRIP=0; do
  envs=""; [ "$v" != "default" ] && envs="$v"
  env $envs timeout 600 python bench.py --steps 20 --warmup 5 --legs ikala --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_g.line 2> $OUT/r06_g.err || tail -n 5 $OUT/r06_g.err
  python - "$v" <<'PY' | tee -a $OUT/r06_g_ikala_ab.txt
import json, sys
d = json.load(open("bench_detail.json"))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        print("%-22s %-8s %.4f ms/clip | %s" % (sys.argv[1], k, L["ms_per_clip"], " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items())))
    elif isinstance(L, dict): print(k, L)
PY
done
: > $OUT/r06_g_stft_timeline.txt
DCS_TRACE_CLIPS=20 DCS_TRACE_TILES=32 DCS_LIB=deepconvsep_amd/_exp_fftwtrace.so timeout 300 python scripts/gpu_fftw_trace.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/r06_g_stft_timeline.txt
DCS_TRACE_CLIPS=1 DCS_TRACE_TILES=4096 DCS_LIB=deepconvsep_amd/_exp_fftwtrace.so timeout 300 python scripts/gpu_fftw_trace.py 2>&1 | grep -v amdgpu.ids | tee -a $OUT/r06_g_stft_timeline.txt
