#!/bin/bash
# Round 5, visit T (final build): the driver's two GPU commands as the driver runs them, smoke(), and the driver's bench command.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1400 python -m pytest tests/ -x -q -m gpu > $OUT/r05_t_pytest_gpu.log 2>&1; echo "pytest exit $?"
grep -v "^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $OUT/r05_t_pytest_gpu.log | tail -n 6 | cut -c1-200
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -n 2
timeout 1200 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_t_driver_stdout.txt 2> $OUT/r05_t_driver.err; echo "driver cmd exit $?"
cp bench_detail.json $OUT/r05_t_bench_detail.json
tail -n 1 $OUT/r05_t_driver_stdout.txt | wc -c
tail -n 1 $OUT/r05_t_driver_stdout.txt | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print({k:d[k] for k in ('metric','value','ms_per_step','whole_path_frac_of_f32_peak','single_stream_ms_per_step')})
print(d['roofline']); print(d['cpu_baseline']['value'], d['parity_check']); print(d['legs_ms_per_clip'])"
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_t_bench_detail.json"))
print(json.dumps(d["cli"]["steady_state"])[:900])
print({k: ((v.get("ms_per_clip"), v.get("whole_path_frac_of_f32_peak")) if isinstance(v, dict) else v) for k, v in d["legs"].items()})
for leg in ("score_informed", "bach10_f32"):
    print(leg, json.dumps(d["legs"][leg]["kernels_ms"]))
    print({k: (v.get("frac"), v.get("traffic_ratio")) for k, v in d["legs"][leg]["kernel_rooflines"].items()})
PY
