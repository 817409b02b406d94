#!/bin/bash
# Round 5, visit P: conv1 of the stride-4 graphs with the input split once on its way into LDS -- parity, then same-box A/B of the legs.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider \
  -k "bach10 or scoreinformed or generic or graph_fixtures or random_draws or channels_last" > $OUT/r05_p_pytest.log 2>&1
echo "pytest exit $?"; tail -n 5 $OUT/r05_p_pytest.log | cut -c1-220
for v in default ${DCS_P_VARIANTS:-}; do
  envs=""; [ "$v" != "default" ] && envs="$v"
  case "$envs" in DCS_LIB=*) envs="DCS_LIB=$PWD/${envs#DCS_LIB=}";; esac
  env $envs timeout 900 python bench.py --steps 20 --warmup 5 --legs ${DCS_P_LEGS:-score_informed,bach10_f32,bach10_f16} --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r05_p.line 2> $OUT/r05_p.err || tail -n 5 $OUT/r05_p.err
  python - "$v" <<'PY'
import json, sys
d = json.load(open("bench_detail.json"))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        print("%-34s %-15s %.4f ms/clip whole %s | %s" % (sys.argv[1][-34:], k, L["ms_per_clip"], L.get("whole_path_frac_of_f32_peak"),
              " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items())))
PY
done
