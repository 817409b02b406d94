#!/bin/bash
# Round 6, visit AA: forward STFT at four waves per SIMD (quarter-circle twiddle table, bins k and M - k from one pair, stores staged in
# the exchange buffer): one resident round for the 20 x 32-tile launch group
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 -p no:cacheprovider -k "stft or transform or dsd or separate or batch or ragged or clips or fused or stereo or variants or guard or launch_shapes" > $OUT/r06_aa_pytest.log 2>&1
echo "pytest exit $?"; tail -n 4 $OUT/r06_aa_pytest.log | cut -c1-200
: > $OUT/r06_aa.txt
for rep in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 4096 > $OUT/r06_aa.line 2> $OUT/r06_aa.err || tail -n 5 $OUT/r06_aa.err
  python - <<'PY' | tee -a $OUT/r06_aa.txt
import json
d = json.load(open("bench_detail.json"))
k = d["launch_group"]["kernels_ms"]
pc = d.get("parity_check") or {}
sat = d.get("saturating") or {}
print("steps 20: ms_per_step %.5f  frac %.4f  parity %s | %s | hbm %s | sat stft %s" % (d["ms_per_step"], d["whole_path_frac_of_f32_peak"], pc.get("ok"), " ".join("%s %.1f" % (a, 1e3 * b) for a, b in k.items()), {a: b.get("frac") for a, b in (d.get("hbm_stages") or {}).items()}, (sat.get("kernels_ms") or {}).get("stft")))
PY
done
timeout 600 python bench.py --legs "" --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_aa.line 2> $OUT/r06_aa.err || tail -n 5 $OUT/r06_aa.err
python - <<'PY' | tee -a $OUT/r06_aa.txt
import json
d = json.load(open("bench_detail.json"))
k = d["launch_group"]["kernels_ms"]
print("default shape: ms_per_step %.5f  frac %.4f | %s | single-stream %.5f" % (d["ms_per_step"], d["whole_path_frac_of_f32_peak"], " ".join("%s %.1f" % (a, 1e3 * b) for a, b in k.items()), d["single_stream_ms_per_step"]))
PY
