#!/bin/bash
# Round 6, visit I: the bottleneck layer of the Bach10 graph on f16 weights under the f16 switch (gemm_f16_longk_kernel), the
# re-grouped GPU suite (fresh-process variants four at a time) with its duration, the legs.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rm -f $OUT/mask_bins.txt $OUT/f16_stats.txt
t0=$SECONDS
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 -p no:cacheprovider --durations=25 > $OUT/r06_i_pytest.log 2>&1
echo "pytest exit $? after $((SECONDS - t0)) s"; tail -n 32 $OUT/r06_i_pytest.log | cut -c1-200
cat $OUT/f16_stats.txt
for i in 1; do
timeout 600 python bench.py --steps 20 --warmup 5 --legs bach10_f16,ikala,score_informed,bach10_f32 --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_i.line 2> $OUT/r06_i.err || tail -n 5 $OUT/r06_i.err
python - <<'PY' | tee -a $OUT/r06_i_legs.txt
import json
d = json.load(open("bench_detail.json"))
print("k20: %.5f ms/step whole %.4f" % (d["ms_per_step"], d["whole_path_frac_of_f32_peak"]))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        print("%-15s %.4f ms/clip | %s" % (k, L["ms_per_clip"], " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items())))
    elif isinstance(L, dict): print(k, L)
PY
done
