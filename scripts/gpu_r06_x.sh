#!/bin/bash
# Round 6, visit X: iKala graph with conv2 + bottleneck layer folded (generic.hip), DCS_FOLD_CONV2 on / off; and the restated
# cross-check of the two DSD kernel families on the adversarial weight sets
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 -p no:cacheprovider -k "ikala or generic or adversarial or random or variants or fallbacks or configs" > $OUT/r06_x_pytest.log 2>&1
echo "pytest exit $?"; tail -n 4 $OUT/r06_x_pytest.log | cut -c1-200
: > $OUT/r06_x_legs.txt
for rep in 1 2 3; do
for v in 1 0; do
DCS_FOLD_CONV2=$v timeout 600 python bench.py --steps 20 --warmup 5 --legs ikala --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_x.line 2> $OUT/r06_x.err || tail -n 5 $OUT/r06_x.err
python - "$v" <<'PY' | tee -a $OUT/r06_x_legs.txt
import json, sys
d = json.load(open("bench_detail.json"))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        pc = L.get("parity_check") or {}
        print("fold=%s %-8s %.4f ms/clip parity %s (pcm %s, net %s) | %s" % (sys.argv[1], k, L["ms_per_clip"], pc.get("ok"), pc.get("pcm_max_err"), pc.get("network_output_max_err"), " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items())))
    elif isinstance(L, dict): print(k, L)
PY
done
done
