#!/bin/bash
# Experiment visit: kernel variants (scripts/build_exp.sh builds) through scripts/gpu_final_exp.py, bench shapes at K = 20.
#   DCS_EXP_LIBS="name ..."  (deepconvsep_amd/_exp_<name>.so, run with the bf16x3 switch on)
set -u
OUT=gpurun_out; mkdir -p $OUT
LOG=$OUT/exp.log; : > $LOG
echo "== f32 (libdcs.so)" | tee -a $LOG
timeout 300 python scripts/gpu_final_exp.py f32 2>&1 | grep -v amdgpu.ids | tee -a $LOG
for n in ${DCS_EXP_LIBS:-}; do
  echo "== $n" | tee -a $LOG
  DCS_EXP_CHECK=${DCS_EXP_CHECK:-1} DCS_LIB=$PWD/deepconvsep_amd/_exp_$n.so DCS_FINAL_BF16X3=1 DCS_FINAL_CBW=2 timeout 300 python scripts/gpu_final_exp.py $n 2>&1 | grep -v amdgpu.ids | tee -a $LOG
done
for v in ${DCS_EXP_BENCH:-}; do   # S<streams>C<clips per launch>K<steps>
  s=${v#S}; s=${s%%C*}; c=${v#*C}; c=${c%%K*}; k=${v#*K}
  echo "== bench --steps $k --streams $s --clips-per-launch $c" | tee -a $LOG
  timeout 300 python bench.py --steps $k --warmup 5 --streams $s --clips-per-launch $c --no-cpu-baseline --legs "" --sat-tiles 0 --no-host-fed > $OUT/exp_bench_$v.json 2> $OUT/exp_bench_$v.err
  python - <<PY | tee -a $LOG
import json
d=json.loads(open("$OUT/exp_bench_$v.json").read().strip().splitlines()[-1])
print("  value %.0f ms/step %.5f rounds %d round_ms %s groups %s" % (d["value"], d["ms_per_step"], d["rounds"], d["round_ms"], d["config"]["launch_groups_per_round"]))
PY
done
