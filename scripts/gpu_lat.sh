#!/bin/bash
# GPU-box visit for the one-batch path: its tests, the stage-by-stage timing sweep, a kernel trace, the launch-floor
# microbenchmark.   gpurun --timeout 1500 -- 'bash scripts/gpu_lat.sh'
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/mask_bins.txt
echo "== pytest one-batch path" | tee $OUT/lat_pytest.log
timeout 900 python -m pytest tests/test_gpu_latency.py -m gpu -q --maxfail=40 --timeout=300 -p no:cacheprovider --durations=8 >> $OUT/lat_pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/lat_pytest.log
tail -n 60 $OUT/lat_pytest.log
if [ "${DCS_LAT_QUICK:-0}" != "1" ]; then
  echo "== pytest whole-path subset (automatic stage selection)" | tee $OUT/lat_pytest2.log
  timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q --maxfail=20 --timeout=300 -p no:cacheprovider \
     -k "dsd_separation or graph_replay or edge_cases or train_auto or separate_batch_equals or randomised or ragged_equals" >> $OUT/lat_pytest2.log 2>&1
  echo "pytest exit $?" | tee -a $OUT/lat_pytest2.log
  tail -n 15 $OUT/lat_pytest2.log
fi
echo "== stage sweep N=2048"
timeout 600 python scripts/gpu_lat_exp.py 2>&1 | tee $OUT/lat_exp_2048.log | tail -n 30
echo "== stage sweep N=1024"
DCS_LAT_EXP_N=1024 timeout 600 python scripts/gpu_lat_exp.py 2>&1 | tee $OUT/lat_exp_1024.log | grep -E "0x00|0xff|HIP-event"
for t in 8 64 128 256; do
  echo "== tiles $t (N=2048): throughput kernels vs one-batch kernels"
  DCS_LAT_MAX_FRAMES=100000 DCS_LAT_EXP_TILES=$t DCS_LAT_EXP_STAGES=0,255 timeout 300 python scripts/gpu_lat_exp.py 2>&1 | tee $OUT/lat_exp_2048_T$t.log | grep -E "^stages|HIP-event"
done
echo "== launch floor"
(cd scripts/ubench && hipcc --offload-arch=gfx950 -O3 launch_floor.hip -o launch_floor 2>/dev/null && timeout 120 ./launch_floor) 2>&1 | tee $OUT/launch_floor.txt
echo "== rocprofv3 kernel trace, one batch per call"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_lat -o lat -- \
   env DCS_LAT_EXP_STAGES=0,255 python $GRAFT_REPO_ROOT/scripts/gpu_lat_exp.py > $GRAFT_REPO_ROOT/$OUT/prof_lat.log 2>&1)
echo "rocprof exit $?"
python scripts/trace_by_grid.py $OUT/prof_lat > $OUT/lat_kernel_durations_by_grid.txt 2>&1
cat $OUT/lat_kernel_durations_by_grid.txt | head -60
python scripts/trace_timeline.py $OUT/prof_lat > $OUT/lat_timeline.txt 2>&1
cat $OUT/lat_timeline.txt
find $OUT/prof_lat -name "*kernel_trace.csv" -delete; find $OUT/prof_lat -name "*.db" -delete
if [ -f deepconvsep_amd/_exp_lattrace.so ]; then
  echo "== in-kernel timeline (s_memtime stamps, experiment build)"
  DCS_LIB=$PWD/deepconvsep_amd/_exp_lattrace.so timeout 300 python scripts/gpu_lat_trace.py 2>&1 | tee $OUT/lat_trace.txt
fi
cat $OUT/mask_bins.txt 2>/dev/null
