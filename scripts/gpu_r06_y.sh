#!/bin/bash
# Round 6, visit Y: conv1 once per FRAME instead of once per tile row on the channels-last graphs (Bach10 f16 / f32-class,
# score-informed) -- same-box A/B against an experiment build without it (deepconvsep_amd/_exp_noperframe.so)
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 -p no:cacheprovider -k "bach10 or scoreinformed or score or generic or configs or channels_last or fallbacks or guard" > $OUT/r06_y_pytest.log 2>&1
echo "pytest exit $?"; tail -n 4 $OUT/r06_y_pytest.log | cut -c1-200
: > $OUT/r06_y_legs.txt
for rep in 1 2; do
for v in new old; do
LIB=""; [ $v = old ] && LIB=deepconvsep_amd/_exp_noperframe.so
DCS_LIB=$LIB timeout 600 python bench.py --steps 20 --warmup 5 --legs bach10_f16,bach10_f32,score_informed --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_y.line 2> $OUT/r06_y.err || tail -n 5 $OUT/r06_y.err
python - "$v" <<'PY' | tee -a $OUT/r06_y_legs.txt
import json, sys
d = json.load(open("bench_detail.json"))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        print("%s %-15s %.4f ms/clip parity %s | %s" % (sys.argv[1], k, L["ms_per_clip"], (L.get("parity_check") or {}).get("ok"), " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items())))
PY
done
done
