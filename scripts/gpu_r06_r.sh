#!/bin/bash
# Round 6, visit R: the lean chained iSTFT (12 waves per workgroup, three per SIMD) against the 8-wave form
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 -p no:cacheprovider -k "dsd or separate or batch or ragged or clips or fused or stereo or istft or stft or inverse or transform or guard" > $OUT/r06_r_pytest.log 2>&1
echo "pytest exit $?"; tail -n 4 $OUT/r06_r_pytest.log | cut -c1-200
: > $OUT/r06_r_istft.txt
for rep in 1 2 3; do
for v in 1 0; do
  DCS_ISTFT_LEAN=$v timeout 600 python bench.py --steps 20 --warmup 5 --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_r.line 2> $OUT/r06_r.err || tail -n 5 $OUT/r06_r.err
  python - "$v" <<'PY' | tee -a $OUT/r06_r_istft.txt
import json, sys
d = json.load(open("bench_detail.json"))
k = d["launch_group"]["kernels_ms"]
pc = d.get("parity_check") or {}
print("DCS_ISTFT_LEAN=%s: ms_per_step %.5f  frac %.4f  istft %.1f us  parity %s pcm %s | %s" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"], 1e3 * k["istft"], pc.get("ok"), pc.get("pcm_max_err", pc.get("max_err")), " ".join("%s %.1f" % (a, 1e3 * b) for a, b in k.items())))
PY
done
done
for v in 1 0; do
  DCS_ISTFT_LEAN=$v timeout 600 python bench.py --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_r.line 2> $OUT/r06_r.err || tail -n 5 $OUT/r06_r.err
  python - "$v" <<'PY' | tee -a $OUT/r06_r_istft.txt
import json, sys
d = json.load(open("bench_detail.json"))
k = d["launch_group"]["kernels_ms"]
print("default shape DCS_ISTFT_LEAN=%s: ms_per_step %.5f  frac %.4f  istft %.1f us single-stream %.5f" % (sys.argv[1], d["ms_per_step"], d["whole_path_frac_of_f32_peak"], 1e3 * k["istft"], d["single_stream_ms_per_step"]))
PY
done
