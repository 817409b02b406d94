#!/bin/bash
# Round 6, visit F: the Bach10 "fp16" config on f16 dense weights with an f16 channels-last D (gemm_f16.hip + the IN16 form of the
# fused decoder) -- tests of the f16 graphs, the pipelined N > 1 schedule of bench.py (world of one / two ranks on one GPU), then
# the legs A/B (default vs DCS_DENSE_F16=0, alternating).
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rm -f $OUT/mask_bins.txt $OUT/f16_stats.txt
timeout 1500 python -m pytest tests -m gpu -q --maxfail=20 --timeout=600 -p no:cacheprovider -k "${DCS_F_K:-bach10 or f16 or channels_last or fused_decoder or variants or gather or two_ranks or guard}" > $OUT/r06_f_pytest.log 2>&1
echo "pytest exit $?"; tail -n 15 $OUT/r06_f_pytest.log | cut -c1-250
cat $OUT/f16_stats.txt 2>/dev/null
: > $OUT/r06_f_legs_ab.txt
for v in default DCS_DENSE_F16=0 default DCS_DENSE_F16=0; do
  envs=""; [ "$v" != "default" ] && envs="$v"
  env $envs timeout 600 python bench.py --steps 20 --warmup 5 --legs bach10_f16 --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_f.line 2> $OUT/r06_f.err || tail -n 5 $OUT/r06_f.err
  python - "$v" <<'PY' | tee -a $OUT/r06_f_legs_ab.txt
import json, sys
d = json.load(open("bench_detail.json"))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        pc = L.get("parity_check") or {}
        print("%-16s %-12s %.4f ms/clip | %s | pcm err %s ok %s" % (sys.argv[1], k, L["ms_per_clip"], " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items()), pc.get("max_abs_pcm_err"), pc.get("ok")))
    elif isinstance(L, dict): print(k, L)
PY
done
