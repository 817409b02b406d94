#!/bin/bash
# Round 6, final visit: the driver's sequence on the final build (pytest -x -m gpu, smoke, the driver's bench command with its
# stdout kept verbatim), the default bench run, then the measurement passes of scripts/gpu_r06_measure.sh.
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
rm -f $OUT/mask_bins.txt $OUT/f16_stats.txt
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0)); import os; print('cpus', os.cpu_count())" > $OUT/env.log 2>&1
grep -m1 "model name" /proc/cpuinfo >> $OUT/env.log
t0=$SECONDS
timeout 1500 python -m pytest tests/ -x -q -m gpu > $OUT/r06_z_pytest_gpu.log 2>&1
echo "pytest exit $? after $((SECONDS - t0)) s"; tail -n 4 $OUT/r06_z_pytest_gpu.log | cut -c1-200
cp $OUT/mask_bins.txt $OUT/r06_z_mask_bins.txt 2>/dev/null; cp $OUT/f16_stats.txt $OUT/r06_z_f16_stats.txt 2>/dev/null
timeout 300 python __graft_entry__.py smoke > $OUT/r06_z_smoke.log 2>&1; echo "smoke exit $?"; tail -n 1 $OUT/r06_z_smoke.log
t0=$SECONDS
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/r06_z_driver_cmd_stdout.txt 2> $OUT/r06_z_driver.err; echo "driver bench exit $? after $((SECONDS - t0)) s"
tail -n 1 $OUT/r06_z_driver_cmd_stdout.txt | cut -c1-2600
cp bench_detail.json $OUT/r06_z_bench_detail_driver_cmd.json
t0=$SECONDS
timeout 900 python bench.py > $OUT/r06_z_default_stdout.txt 2> $OUT/r06_z_default.err; echo "default bench exit $? after $((SECONDS - t0)) s"
tail -n 1 $OUT/r06_z_default_stdout.txt | cut -c1-1200
cp bench_detail.json $OUT/r06_z_bench_detail_default.json
bash scripts/gpu_r06_measure.sh 2>&1 | tail -n 45
