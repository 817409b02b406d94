#!/bin/bash
# Round 6, measurement visit of the FINAL build: (1) rocprofv3 --kernel-trace --stats of the driver's command and of the legs
# -> kernel durations by grid; (2) FETCH_SIZE / WRITE_SIZE passes (separate rocprofv3 --pmc runs, --kernel-trace only) of the DSD
# launch shapes and of every leg -> gpurun_out/traffic.json (-> profiles/r06_traffic.json, read by bench.py); (3) wave / wait /
# instruction-mix passes at the driver's shape (20 x 32 tiles, one stream) -> counters_summary_k20.txt.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
rm -rf $OUT/pmc_* $OUT/prof_k20 $OUT/prof_legs
B="python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-host-fed --no-cli --no-parity-check --min-time 0.02 --max-rounds 6"
cd /tmp
echo "== (1) kernel traces"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_k20 -o bench -- python $GRAFT_REPO_ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --legs= --sat-tiles 0 --no-cpu-baseline --no-host-fed --no-cli > $OUT/prof_k20.json 2> $OUT/prof_k20.err; echo "trace k20 exit $?"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_legs -o bench -- $B --only-legs --legs ikala,bach10_f16,bach10_f32,score_informed > $OUT/prof_legs.json 2> $OUT/prof_legs.err; echo "trace legs exit $?"
(cd $GRAFT_REPO_ROOT && python scripts/trace_by_grid.py $OUT/prof_k20 > $OUT/r06_kernel_durations_by_grid_k20.txt 2>&1; python scripts/trace_by_grid.py $OUT/prof_legs > $OUT/r06_kernel_durations_by_grid_legs.txt 2>&1)
for f in $(find $OUT/prof_k20 -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/r06_k20_kernel_stats.csv; head -n 12 $f; done
echo "== (2) traffic"
run() { tag=$1; shift
  for c in fetch:FETCH_SIZE write:WRITE_SIZE; do
    n=${c%%:*}; ctr=${c##*:}
    timeout 600 rocprofv3 --kernel-trace --pmc $ctr --output-format csv -d $OUT/pmc_${n}_$tag -o p -- "$@" > $OUT/pmc_${n}_$tag.json 2> $OUT/pmc_${n}_$tag.err
    echo "pmc $n $tag exit $?"
  done; }
run k20 $B --steps 20 --warmup 5 --legs= --sat-tiles 0
run g32 $B --steps 32 --warmup 8 --streams 1 --legs= --sat-tiles 4096
for leg in ikala bach10_f16 bach10_f32 score_informed; do
  run leg_$leg $B --only-legs --legs $leg
done
cd $GRAFT_REPO_ROOT
python scripts/traffic_summary.py $OUT | tee $OUT/r06_traffic_summary.txt | tail -70
echo "== (3) counters at the driver's shape"
cd /tmp
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --streams 1 --no-cpu-baseline --no-host-fed --no-cli --no-parity-check --legs= --sat-tiles 0 --min-time 0.02 --max-rounds 6"
runc() { name=$1; shift
  rm -rf $OUT/pmc_$name
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- $CMD > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pmc $name exit $?"; }
runc waves GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
runc insts GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT
cd $GRAFT_REPO_ROOT
cp $OUT/traffic.json $OUT/r06_traffic.json     # (counters_summary.py writes a traffic.json of its own)
python scripts/counters_summary.py $OUT > $OUT/r06_counters_summary_k20.txt 2>&1
grep -v "^lat_" $OUT/r06_counters_summary_k20.txt | head -40
find $OUT -name "*.db" -delete; find $OUT -path "*pmc_*" -name "*kernel_trace.csv" -delete; find $OUT -path "*prof_*" -name "*kernel_trace.csv" -delete
rm -rf $OUT/pmc_waves $OUT/pmc_insts
du -sh $OUT | tail -1
