#!/bin/bash
# PMC passes (each its own rocprofv3 run, --kernel-trace only -- never combined with other trace domains):
#   FETCH_SIZE, WRITE_SIZE            -> gpurun_out/traffic.json          (profiles/rNN_traffic.json, read by bench.py)
#   wave / wait / instruction mix     -> gpurun_out/counters_summary.txt  (profiles/rNN_pmc_summary.txt)
# Workload: bench.py on ONE stream (kernel durations are then not time-sliced with another stream's kernels), 16 x 32-tile
# launch groups, the 4096-tile clip and the 32-tile batch.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 16 --streams 1 --no-cpu-baseline --no-host-fed --no-cli --no-parity-check --legs=${DCS_COUNTER_LEGS:-} --sat-tiles 4096 --min-time 0.02 --max-rounds 6"
cd /tmp
run() { name=$1; shift
  rm -rf $OUT/pmc_$name
  timeout 900 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- $CMD > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pmc $name exit $?"; }
run fetch FETCH_SIZE
run write WRITE_SIZE
run waves GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run insts GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
cd $GRAFT_REPO_ROOT
python scripts/counters_summary.py $OUT > $OUT/counters_summary.txt 2>&1
cat $OUT/counters_summary.txt | head -80
