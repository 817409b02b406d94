#!/bin/bash
# A/B of environment switches on the DSD bench shapes without legs / CLI / CPU baseline:
#   DCS_AB_K="stft or istft" (pytest -k expression, empty = skip)   DCS_AB_VARIANTS="default NAME=VAL ..."
set -u
OUT=gpurun_out; mkdir -p $OUT
if [ -n "${DCS_AB_K:-}" ]; then
  timeout 1200 python -m pytest tests -m gpu -q -x --timeout=240 -p no:cacheprovider -k "$DCS_AB_K" > $OUT/ab_pytest.log 2>&1; echo "pytest exit $?"; tail -n 6 $OUT/ab_pytest.log
fi
for v in ${DCS_AB_VARIANTS:-default}; do
  envs=""; [ "$v" != "default" ] && envs="$v"
  vn=${v//\//_}
  env $envs timeout 300 python bench.py --steps 384 --warmup 40 --no-cpu-baseline --no-host-fed --no-cli --no-parity-check --legs= > $OUT/ab_$vn.json 2> $OUT/ab_$vn.err; echo "== $v exit $?"; tail -n 2 $OUT/ab_$vn.err
  python - <<PY
import json
d=json.load(open("$OUT/ab_$vn.json"))
print("headline %.0f frames/s, %.5f ms/step" % (d['value'], d['ms_per_step']))
print("single   %.5f ms/step" % d['single_stream']['ms_per_step'], d['single_stream']['kernels_ms'])
g=d['launch_group']; print("group    sum %.4f" % g['kernels_ms_sum'], g['kernels_ms'])
s=d.get('saturating')
if s: print("sat      %.4f ms" % s['ms_per_step'], s['kernels_ms'])
PY
done
