#!/bin/bash
# A/B visit for the generic-graph legs: a few parity tests, then `bench.py --legs ...` once per environment variant.
#   DCS_AB_TESTS="pytest -k expression"   DCS_AB_LEGS="bach10_f16,score_informed"   DCS_AB_VARIANTS="NAME=V;NAME2=V2 ..."
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/legs_ab.log; : > $LOG
if [ -n "${DCS_AB_TESTS:-}" ]; then
  timeout 900 python -m pytest tests -m gpu -x -q -k "$DCS_AB_TESTS" 2>&1 | grep -v amdgpu.ids | tail -15 | tee -a $LOG
fi
i=0
for v in "" ${DCS_AB_VARIANTS:-}; do
  echo "== variant '$v'" | tee -a $LOG
  env $(echo "$v" | tr ';' ' ') timeout 600 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --sat-tiles 0 --no-host-fed \
      --legs "${DCS_AB_LEGS:-bach10_f16,score_informed}" > $OUT/legs_ab_$i.json 2> $OUT/legs_ab_$i.err
  python - <<PY | tee -a $LOG
import json
d=json.loads(open("$OUT/legs_ab_$i.json").read().strip().splitlines()[-1])
for k,v in d.get("legs",{}).items():
    print("  %-15s %.4f ms/clip  %s" % (k, v["ms_per_clip"], {a:round(b,4) for a,b in sorted(v["kernels_ms"].items(), key=lambda x:-x[1])}))
PY
  i=$((i+1))
done
