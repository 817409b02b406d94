#!/bin/bash
# Round 5, visit F: the tests touched since the last full run, the round-5 traffic passes, the driver's command.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider \
  -k "red_zones or dcs_gather_through or batch_driver or pcm16 or pcm_to_int16 or compute_transform or reference_graph_fixtures or guard" > $OUT/r05_f_pytest.log 2>&1
echo "pytest exit $?"; tail -n 6 $OUT/r05_f_pytest.log | cut -c1-200
bash scripts/gpu_traffic_r05.sh 2>&1 | tail -45
cp profiles/r04_traffic.json /tmp/r05_traffic.json; python scripts/traffic_merge.py $OUT/traffic.json /tmp/r05_traffic.json; cp /tmp/r05_traffic.json profiles/r05_traffic.json; cp /tmp/r05_traffic.json $OUT/r05_traffic.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_f_driver_stdout.txt 2> $OUT/r05_f_driver.err; echo "driver cmd exit $?"
cp bench_detail.json $OUT/r05_f_bench_detail.json
tail -c 2600 $OUT/r05_f_driver_stdout.txt; echo
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_f_bench_detail.json"))
print(json.dumps(d["launch_group"], indent=None)[:1500])
print(json.dumps(d["cli"]["steady_state"], indent=None))
print({k: (v.get("ms_per_clip") if isinstance(v, dict) else v) for k, v in d["legs"].items()})
print(json.dumps(d["legs"]["transform"]["cases"]["N1024_float64"]))
PY
