#!/bin/bash
# Round 5, visit B: the new score-informed / single-branch / mask-bin tests, the random draws, then a short bench with parity.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT; rm -f $OUT/mask_bins.txt
timeout 1500 python -m pytest tests -m gpu -q -x --timeout=600 -p no:cacheprovider \
  -k "${DCS_B_K:-score or random or single_branch or ikala_trainer or generic_graphs or dense_layers or chunk}" > $OUT/r05_b_pytest.log 2>&1
echo "pytest exit $?"; tail -n 12 $OUT/r05_b_pytest.log
cp $OUT/mask_bins.txt $OUT/r05_b_mask_bins.txt 2>/dev/null
timeout 600 python bench.py --steps 20 --warmup 5 --legs ikala --no-cli --no-host-fed --sat-tiles 0 > $OUT/r05_b_bench.line 2> $OUT/r05_b_bench.err; echo "bench exit $?"; tail -n 3 $OUT/r05_b_bench.err
cp bench_detail.json $OUT/r05_b_bench_detail.json
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_b_bench_detail.json"))
print(json.dumps(d["parity_check"])[:1500])
print(json.dumps(d["legs"]["ikala"].get("parity_check"))[:1500])
PY
