#!/bin/bash
# Round 5, visit O (final build): smoke(), the driver's command (stdout + detail kept), a rocprofv3 kernel trace of that command
# summarised by grid.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -n 3
t0=$(date +%s.%N)
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_o_driver_stdout.txt 2> $OUT/r05_o_driver.err; echo "driver cmd exit $? in $(echo "$(date +%s.%N) - $t0" | bc) s"
cp bench_detail.json $OUT/r05_o_bench_detail.json
tail -n 1 $OUT/r05_o_driver_stdout.txt | wc -c
tail -n 1 $OUT/r05_o_driver_stdout.txt | cut -c1-2600; echo
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_o_bench_detail.json"))
print(json.dumps(d["launch_group"], indent=None)[:1800])
print(json.dumps(d["cli"]["steady_state"], indent=None))
print({k: ((v.get("ms_per_clip"), v.get("whole_path_frac_of_f32_peak")) if isinstance(v, dict) else v) for k, v in d["legs"].items()})
print(json.dumps(d["legs"]["score_informed"].get("roofline"))[:900])
PY
DCS_AB_VARIANTS="default default" DCS_K20_REPS=1 DCS_K20_TRACE=1 bash scripts/gpu_k20_ab.sh
