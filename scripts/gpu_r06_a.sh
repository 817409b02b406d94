#!/bin/bash
# Round 6, visit A: the new mask criterion + colconv_fwd_x3_kernel as the default conv2 of the Bach10 / score-informed graphs
# + pad-column zeroing instead of whole-buffer memsets.  Full GPU suite, smoke, the driver's bench command, then the legs A/B
# (default vs DCS_CONV2_X3=0, alternating) and the 24 random draws with BOTH conv2 kernels (mask-bin reports side by side).
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rm -f $OUT/mask_bins.txt
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0)); import os; print('cpus', os.cpu_count())" > $OUT/env.log 2>&1
echo "== pytest -m gpu" | tee $OUT/pytest.log
timeout 1500 python -m pytest tests -m gpu -q --maxfail=60 --timeout=400 -p no:cacheprovider --durations=15 >> $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest.log
tail -n 40 $OUT/pytest.log | cut -c1-220
cp $OUT/mask_bins.txt $OUT/r06_a_mask_bins_default.txt 2>/dev/null
echo "== smoke"
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?"; tail -n 2 $OUT/smoke.log
echo "== random draws with slabconv_ps_kernel (DCS_CONV2_X3=0)"
rm -f $OUT/mask_bins.txt
DCS_CONV2_X3=0 timeout 600 python -m pytest tests/test_gpu_random.py -m gpu -q -p no:cacheprovider > $OUT/r06_a_random_x3off.log 2>&1; echo "exit $?"; tail -n 3 $OUT/r06_a_random_x3off.log
cp $OUT/mask_bins.txt $OUT/r06_a_mask_bins_x3off.txt 2>/dev/null
echo "== the driver's command"
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_a_driver_stdout.txt 2> $OUT/r06_a_driver.err; echo "bench exit $?"; tail -n 3 $OUT/r06_a_driver.err
tail -n 1 $OUT/r06_a_driver_stdout.txt | cut -c1-3000
cp bench_detail.json $OUT/r06_a_bench_detail.json
echo "== legs A/B"
for v in default DCS_CONV2_X3=0 default DCS_CONV2_X3=0; do
  envs=""; [ "$v" != "default" ] && envs="$v"
  env $envs timeout 600 python bench.py --steps 20 --warmup 5 --legs score_informed,bach10_f32,bach10_f16 --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_a.line 2> $OUT/r06_a.err || tail -n 5 $OUT/r06_a.err
  python - "$v" <<'PY' | tee -a $OUT/r06_a_legs_ab.txt
import json, sys
d = json.load(open("bench_detail.json"))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        print("%-16s %-15s %.4f ms/clip whole %s | %s" % (sys.argv[1], k, L["ms_per_clip"], L.get("whole_path_frac_of_f32_peak"), " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items())))
PY
done
