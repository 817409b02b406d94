#!/bin/bash
# Round 5, visit Q: decoder stage 1 with both rows of a pair interleaved (four accumulator chains) -- parity, then same-box A/B.
set -u
export TMPDIR=/tmp

OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider \
  -k "decoder or channels_last or bach10_fused or scoreinformed_batch or bach10_full_size or graph_fixtures" > $OUT/r05_q_pytest.log 2>&1
echo "pytest exit $?"; tail -n 5 $OUT/r05_q_pytest.log | cut -c1-220
for v in default ${DCS_Q_VARIANTS:-}; do
  envs=""; [ "$v" != "default" ] && envs="$v"
  case "$envs" in DCS_LIB=*) envs="DCS_LIB=$PWD/${envs#DCS_LIB=}";; esac
  env $envs timeout 900 python bench.py --steps 20 --warmup 5 --legs ${DCS_Q_LEGS:-score_informed,bach10_f32} --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r05_q.line 2> $OUT/r05_q.err || tail -n 5 $OUT/r05_q.err
  python - "$v" <<'PY'
import json, sys
d = json.load(open("bench_detail.json"))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        print("%-34s %-15s %.4f ms/clip whole %s | %s" % (sys.argv[1][-34:], k, L["ms_per_clip"], L.get("whole_path_frac_of_f32_peak"),
              " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items())))
PY
done
