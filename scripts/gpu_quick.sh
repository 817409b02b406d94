#!/bin/bash
# Quick GPU visit: parity tests + a few bench variants (no rocprof).  Output in gpurun_out/quick_*.
set -u
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 --timeout=240 -p no:cacheprovider > $OUT/quick_pytest.log 2>&1; echo "pytest exit $?"; tail -n 15 $OUT/quick_pytest.log
for v in ${DCS_VARIANTS:-default}; do
  case $v in
    default) envs="";;
    lds64) envs="DCS_ISTFT_LDS_KB=64";;
    lds36) envs="DCS_ISTFT_LDS_KB=36";;
    *) envs="$v";;
  esac
  echo "== bench variant $v ($envs)"
  env $envs timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu-baseline > $OUT/quick_bench_$v.json 2> $OUT/quick_bench_$v.err; echo "exit $?"
  python - <<PY
import json
d=json.load(open("$OUT/quick_bench_$v.json"))
print("32t: value %.0f ms/step %.4f frac %.4f" % (d['value'], d['ms_per_step'], d['roofline']['frac']), d['kernels_ms'])
s=d['saturating']; print("SAT: value %.0f ms/step %.4f frac %.4f" % (s['value'], s['ms_per_step'], s['roofline']['frac']), s['kernels_ms'])
PY
done
