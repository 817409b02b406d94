#!/bin/bash
# Round 4: PMC passes at the DRIVER's launch shape (bench.py --steps 20, one stream): wave / wait / instruction mix of every
# kernel of the 640-tile launch group (counters_summary.py) -- each pass its own rocprofv3 run with --kernel-trace only.
#   DCS_PMC_TRAFFIC=1 adds the FETCH_SIZE / WRITE_SIZE passes;  DCS_PMC_ENV="NAME=VAL ..." environment of the workload
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
CMD="env ${DCS_PMC_ENV:-} python $GRAFT_REPO_ROOT/bench.py --steps ${DCS_PMC_STEPS:-20} --warmup 5 --streams 1 --no-cpu-baseline --no-host-fed --no-cli --no-parity-check --legs= --sat-tiles ${DCS_PMC_SAT:-0} --min-time 0.02 --max-rounds 6"
cd /tmp
run() { name=$1; shift
  rm -rf $OUT/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- $CMD > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pmc $name exit $?"; }
if [ "${DCS_PMC_TRAFFIC:-0}" = "1" ]; then
  run fetch FETCH_SIZE
  run write WRITE_SIZE
fi
run waves GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run insts GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
cd $GRAFT_REPO_ROOT
python scripts/counters_summary.py $OUT > $OUT/counters_summary_k20.txt 2>&1
grep -v "^lat_" $OUT/counters_summary_k20.txt | head -60
rm -rf $OUT/pmc_waves $OUT/pmc_insts
