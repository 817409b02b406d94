#!/bin/bash
# Round 5, visit K: the score-informed graph's two InverseLayers through the fused bf16 x 3 decoder (four output channels).
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 2400 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider \
  -k "score or bach10 or single_branch or generic or graph_fixtures or random or decoder or x3" > $OUT/r05_k_pytest.log 2>&1
echo "pytest exit $?"; tail -n 12 $OUT/r05_k_pytest.log | cut -c1-220
for v in default DCS_DECODER_X3=0; do
  envs=""; [ "$v" != "default" ] && envs="$v"
  env $envs timeout 900 python bench.py --steps 20 --warmup 5 --legs score_informed,bach10_f32 --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r05_k_$v.line 2> $OUT/r05_k_$v.err || tail -n 5 $OUT/r05_k_$v.err
  cp bench_detail.json $OUT/r05_k_$v.json
  python - "$v" <<'PY'
import json, sys
d = json.load(open("bench_detail.json"))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        print("%-18s %-16s %.4f ms/clip whole %.4f | %s | parity %s" % (sys.argv[1], k, L["ms_per_clip"], L.get("whole_path_frac_of_f32_peak") or 0,
              " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items()), (L.get("parity_check") or {}).get("max_abs_pcm_err")))
    else:
        print(sys.argv[1], k, str(L)[:300])
PY
done
