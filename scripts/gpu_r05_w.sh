#!/bin/bash
# Round 5, visit W: conv2 of the Bach10 / score-informed graphs with the weights in registers (opt-in since: DCS_CONV2_X3=1 / unset; the script predates that and compares default-on with DCS_CONV2_X3=0).
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --timeout=600 -p no:cacheprovider -k "${DCS_W_K:-generic_graphs_match or bach10_full_size or scoreinformed_batch or fused_decoders_at_full}" > $OUT/r05_w_pytest.log 2>&1
echo "pytest exit $?"; tail -n 4 $OUT/r05_w_pytest.log | cut -c1-200
for v in default DCS_CONV2_X3=0 default DCS_CONV2_X3=0; do
  envs=""; [ "$v" != "default" ] && envs="$v"
  env $envs timeout 600 python bench.py --steps 20 --warmup 5 --legs score_informed,bach10_f32 --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r05_w.line 2> $OUT/r05_w.err || tail -n 5 $OUT/r05_w.err
  python - "$v" <<'PY'
import json, sys
d = json.load(open("bench_detail.json"))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        print("%-16s %-15s %.4f ms/clip whole %s | %s" % (sys.argv[1], k, L["ms_per_clip"], L.get("whole_path_frac_of_f32_peak"), " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items())))
PY
done
