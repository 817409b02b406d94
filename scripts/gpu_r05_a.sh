#!/bin/bash
# Round 5, visit A: the driver's exact command (is the last stdout line parseable?), then the k20 A/B of STFT residency variants.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r05_a_driver_stdout.txt 2> $OUT/r05_a_driver.err; echo "driver cmd exit $?"
cp bench_detail.json $OUT/r05_a_bench_detail.json
wc -c $OUT/r05_a_driver_stdout.txt; tail -n 1 $OUT/r05_a_driver_stdout.txt | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(len(json.dumps(d)), d['value'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['parity_check'])"
bash scripts/gpu_k20_ab.sh
