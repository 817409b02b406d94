#!/bin/bash
# Round 6, visit N: Bach10 f16 switch -- conv1 hands its output to the f16 conv2 as f16, channels-last, 32 channels per position.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rm -f $OUT/mask_bins.txt $OUT/f16_stats.txt
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 -p no:cacheprovider -k "bach10 or f16 or channels_last or fused_decoder or variants or guard" > $OUT/r06_n_pytest.log 2>&1
echo "pytest exit $?"; tail -n 4 $OUT/r06_n_pytest.log | cut -c1-200; cat $OUT/f16_stats.txt
for i in 1 2 3; do
timeout 600 python bench.py --steps 20 --warmup 5 --legs bach10_f16 --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_n.line 2> $OUT/r06_n.err || tail -n 5 $OUT/r06_n.err
python - <<'PY' | tee -a $OUT/r06_n_legs.txt
import json
d = json.load(open("bench_detail.json"))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        print("%-15s %.4f ms/clip | %s" % (k, L["ms_per_clip"], " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items())))
    elif isinstance(L, dict): print(k, L)
PY
done
