#!/bin/bash
# Quick GPU visit: parity tests + bench variants (no rocprof).  Output in gpurun_out/quick_*.
#   DCS_SWEEP=1: B<b>S<s> variants run b*s*8 steps (whole launch groups only) and skip the saturating leg
#   DCS_VARIANTS="default S8 B16 B8S3 ENV=VAL ..."   (S<k> = --streams k; B<b> = --clips-per-launch b; NAME=VAL exported)
set -u
OUT=gpurun_out; mkdir -p $OUT
if [ "${DCS_SKIP_TESTS:-0}" != "1" ]; then
timeout 900 python -m pytest tests -m gpu -q --maxfail=20 --timeout=240 -p no:cacheprovider > $OUT/quick_pytest.log 2>&1; echo "pytest exit $?"; tail -n 15 $OUT/quick_pytest.log
fi
for v in ${DCS_VARIANTS:-default}; do
  envs=""; extra=""; steps=400; sat=""; vn=${v//\//_}
  case $v in
    default) ;;
    B[0-9]*S[0-9]*) b=${v#B}; b=${b%%S*}; k=${v##*S}; extra="--clips-per-launch $b --streams $k"
                    if [ -n "${DCS_SWEEP:-}" ]; then steps=$((b*k*8)); sat="--sat-tiles 0"; fi;;
    S[0-9]*) extra="--streams ${v#S}";;
    B[0-9]*) extra="--clips-per-launch ${v#B}";;
    *) envs="$v";;
  esac
  echo "== bench variant $v ($envs $extra)"
  env $envs timeout 600 python bench.py --steps $steps --warmup 40 --no-cpu-baseline $sat $extra > $OUT/quick_bench_$vn.json 2> $OUT/quick_bench_$vn.err; echo "exit $?"; tail -n 3 $OUT/quick_bench_$vn.err
  python - <<PY
import json
d=json.load(open("$OUT/quick_bench_$vn.json"))
print("32t x%d clips/launch x%d streams: value %.0f ms/step %.4f frac %.4f (final %.4f ms)" % (d['config']['clips_per_launch'], d['config']['streams_per_gpu'], d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['avg_kernel_ms']))
s1=d['single_stream']; print("32t single: value %.0f ms/step %.4f frac %.4f" % (s1['value'], s1['ms_per_step'], s1['roofline']['frac']), s1['kernels_ms'])
g=d['launch_group']; print("GROUP %d clips: sum %.4f" % (g['clips'], g['kernels_ms_sum']), g['kernels_ms'])
h=d.get('host_fed')
if h: print("HOST-FED: value %.0f ms/step %.4f PCIe %.1f GB/s" % (h['value'], h['ms_per_step'], h['pcie_GBps']))
s=d.get('saturating')
if s: print("SAT: value %.0f ms/step %.4f frac %.4f" % (s['value'], s['ms_per_step'], s['roofline']['frac']), s['kernels_ms'])
PY
done
