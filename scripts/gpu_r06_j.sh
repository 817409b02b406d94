#!/bin/bash
# Round 6, visit J: conv2 of the Bach10 graph writing f16 under the f16 switch, the bottleneck layer multiplying those rows --
# f16 / Bach10 tests + guard harness, then the leg twice.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rm -f $OUT/mask_bins.txt $OUT/f16_stats.txt
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 -p no:cacheprovider -k "bach10 or f16 or channels_last or fused_decoder or variants or guard or scoreinformed" > $OUT/r06_j_pytest.log 2>&1
echo "pytest exit $?"; tail -n 5 $OUT/r06_j_pytest.log | cut -c1-200; cat $OUT/f16_stats.txt
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --legs bach10_f16 --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_j.line 2> $OUT/r06_j.err || tail -n 5 $OUT/r06_j.err
python - <<'PY' | tee -a $OUT/r06_j_legs.txt
import json
d = json.load(open("bench_detail.json"))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        print("%-15s %.4f ms/clip | %s" % (k, L["ms_per_clip"], " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items())))
    elif isinstance(L, dict): print(k, L)
PY
done
