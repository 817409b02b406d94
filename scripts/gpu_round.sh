#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench, rocprof kernel stats.  Run through gpurun:
#   gpurun --timeout 1500 -- 'bash scripts/gpu_round.sh'
# Everything worth keeping lands in gpurun_out/ (merged back into the repo copy).
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0)); import os; print('cpus', os.cpu_count())" > $OUT/env.log 2>&1
echo "== pytest -m gpu" | tee $OUT/pytest.log
timeout 1200 python -m pytest tests -m gpu -q --maxfail=60 --timeout=240 -p no:cacheprovider >> $OUT/pytest.log 2>&1
echo "pytest exit $?" | tee -a $OUT/pytest.log
tail -n 60 $OUT/pytest.log
echo "== smoke"
timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -n 5 $OUT/smoke.log
echo "== bench"
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; cat $OUT/bench.json; tail -n 5 $OUT/bench.err
if [ "${DCS_PROFILE:-1}" = "1" ]; then
  echo "== rocprofv3 kernel stats"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 160 --warmup 32 --no-cpu-baseline --no-host-fed > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
  echo "rocprof exit $?"
  find $OUT/prof -name "*stats*" | head; for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -n 25 $f; done
fi
