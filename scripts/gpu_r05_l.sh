#!/bin/bash
# Round 5, visit L: score-informed fused decoder with the per-channel tails software-pipelined.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1200 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider \
  -k "scoreinformed or channels_last_dense or single_branch or sum_normalised" > $OUT/r05_l_pytest.log 2>&1
echo "pytest exit $?"; tail -n 6 $OUT/r05_l_pytest.log | cut -c1-220
for v in default ${DCS_L_VARIANTS:-}; do
  envs=""; [ "$v" != "default" ] && envs="$v"
  case "$envs" in DCS_LIB=*) envs="DCS_LIB=$PWD/${envs#DCS_LIB=}";; esac
  env $envs timeout 900 python bench.py --steps 20 --warmup 5 --legs score_informed --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r05_l.line 2> $OUT/r05_l.err || tail -n 5 $OUT/r05_l.err
  python - "$v" <<'PY'
import json, sys
d = json.load(open("bench_detail.json"))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        print("%-28s %-16s %.4f ms/clip whole %.4f | %s" % (sys.argv[1], k, L["ms_per_clip"], L.get("whole_path_frac_of_f32_peak") or 0,
              " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items())))
PY
done
