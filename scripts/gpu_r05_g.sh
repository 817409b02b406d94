#!/bin/bash
# Round 5, visit G: the Nyquist-bin path of the final kernel -- parity tests, A/B against the previous object on the same box
# (alternating), traffic passes at the new grid, the driver's command.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -q --timeout=900 -p no:cacheprovider \
  -k "bf16x3 or bench_launch_shapes or separate_batch_equals or separate_ragged or red_zones or batch_driver or kernel_variants or compute_transform or bench_two_ranks_on_this_gpu" > $OUT/r05_g_pytest.log 2>&1
echo "pytest exit $?"; tail -n 8 $OUT/r05_g_pytest.log | cut -c1-200
DCS_AB_VARIANTS="default DCS_LIB=deepconvsep_amd/_exp_nonyq.so default DCS_LIB=deepconvsep_amd/_exp_nonyq.so" DCS_K20_REPS=1 DCS_K20_TRACE=1 bash scripts/gpu_k20_ab.sh
bash scripts/gpu_traffic_r05.sh 2>&1 | grep -E "exit|final_bf16x3|istft_chain|stft_forward"
cp profiles/r05_traffic.json /tmp/r05_traffic.json; python scripts/traffic_merge.py $OUT/traffic.json /tmp/r05_traffic.json; cp /tmp/r05_traffic.json profiles/r05_traffic.json; cp /tmp/r05_traffic.json $OUT/r05_traffic.json
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_g_driver_stdout.txt 2> $OUT/r05_g_driver.err; echo "driver cmd exit $?"
cp bench_detail.json $OUT/r05_g_bench_detail.json
tail -c 2600 $OUT/r05_g_driver_stdout.txt; echo
python - <<'PY'
import json
d = json.load(open("gpurun_out/r05_g_bench_detail.json"))
print(json.dumps(d["launch_group"], indent=None)[:1800])
print(json.dumps(d["cli"]["steady_state"], indent=None))
print(json.dumps(d["legs"]["transform"]["cases"]["N1024_float64"]))
PY
