#!/bin/bash
# One GPU-box visit: parity tests, smoke, bench (default and as the driver runs it), rocprof kernel stats.
#   gpurun --timeout 1800 -- 'bash scripts/gpu_round.sh'
# Everything worth keeping lands in gpurun_out/ (merged back into the repo copy).
#   DCS_SKIP_TESTS=1 skips pytest; DCS_PROFILE=0 skips rocprof; DCS_BENCH_ARGS="..." extra arguments of the default bench run
set -u
export TMPDIR=/tmp
OUT=gpurun_out
mkdir -p $OUT
rm -f $OUT/mask_bins.txt $OUT/f16_stats.txt
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0)); import os; print('cpus', os.cpu_count())" > $OUT/env.log 2>&1
grep -m1 "model name" /proc/cpuinfo >> $OUT/env.log
if [ "${DCS_SKIP_TESTS:-0}" != "1" ]; then
  echo "== pytest -m gpu" | tee $OUT/pytest.log
  timeout 1500 python -m pytest tests -m gpu -q --maxfail=60 --timeout=400 -p no:cacheprovider --durations=15 >> $OUT/pytest.log 2>&1
  echo "pytest exit $?" | tee -a $OUT/pytest.log
  tail -n 70 $OUT/pytest.log
  echo "== smoke"
  timeout 300 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1; echo "smoke exit $?" | tee -a $OUT/smoke.log; tail -n 5 $OUT/smoke.log
fi
echo "== bench (default)"
timeout 900 python bench.py ${DCS_BENCH_ARGS:-} > $OUT/bench.json 2> $OUT/bench.err; echo "bench exit $?"; tail -n 5 $OUT/bench.err
echo "== bench as the driver runs it (--steps 20 --warmup 5)"
timeout 600 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --legs "" --sat-tiles 0 --no-host-fed > $OUT/bench_k20.json 2> $OUT/bench_k20.err; echo "bench k20 exit $?"; tail -n 3 $OUT/bench_k20.err
python - <<'PY'
import json
for f in ("gpurun_out/bench.json", "gpurun_out/bench_k20.json"):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, "value %.0f frames/s, ms/step %.5f, rounds %d, round_ms %s" % (d["value"], d["ms_per_step"], d["rounds"], d["round_ms"]))
    print("  roofline", {k: d["roofline"][k] for k in ("achieved", "frac", "avg_kernel_ms", "launches", "tiles_per_launch")})
    s1 = d["single_stream"]; print("  single: ms/step %.5f" % s1["ms_per_step"], s1["kernels_ms"], "frac", s1["roofline"]["frac"])
    g = d["launch_group"]; print("  group of %d: sum %.4f" % (g["clips"], g["kernels_ms_sum"]), g["kernels_ms"])
    if d.get("saturating"): s = d["saturating"]; print("  SAT: %.0f frames/s, %.4f ms, final frac %.4f, whole path %.1f TF" % (s["value"], s["ms_per_step"], s["roofline"]["frac"], s["whole_path_algorithmic_tflops"]), s["kernels_ms"])
    if d.get("cpu_baseline"): print("  CPU", {k: d["cpu_baseline"][k] for k in ("value", "cores", "single_thread", "all_cores", "cpu_model")})
    if d.get("host_fed"): print("  host-fed %.0f frames/s" % d["host_fed"]["value"])
    for k, v in (d.get("legs") or {}).items():
        if "error" in v: print("  LEG", k, "ERROR", v["error"]); continue
        print("  LEG %s: %.3f ms per clip (%d tiles), %.0f frames/s, x%.0f RT, whole path %.1f TF" % (k, v["ms_per_clip"], v["tiles"], v["value"], v["x_realtime"], v["whole_path_algorithmic_tflops"]))
        print("     kernels", v["kernels_ms"]); print("     rooflines", v["kernel_rooflines"])
        print("     dominant", {kk: v["roofline"][kk] for kk in ("tag", "bound", "achieved", "unit", "frac", "avg_kernel_ms", "tiles_per_launch")})
        if v.get("cpu_baseline"): print("     CPU", {kk: v["cpu_baseline"][kk] for kk in ("value", "cores", "single_thread")})
PY
if [ "${DCS_PROFILE:-1}" = "1" ]; then
  echo "== rocprofv3 kernel stats"
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof -o bench -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 160 --warmup 32 --no-cpu-baseline --no-host-fed > $GRAFT_REPO_ROOT/$OUT/prof_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof.err)
  echo "rocprof exit $?"
  for f in $(find $OUT/prof -name "*kernel_stats*.csv" | head -1); do head -n 14 $f; done
  python scripts/trace_by_grid.py $OUT/prof > $OUT/kernel_durations_by_grid.txt 2>&1
  # the same on ONE stream: kernel durations that are not time-sliced with another stream's kernels (what the HIP events of
  # bench.py's event rounds measure)
  (cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof1 -o bench -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 160 --warmup 32 --streams 1 --no-cpu-baseline --no-host-fed --legs= > $GRAFT_REPO_ROOT/$OUT/prof1_bench.json 2> $GRAFT_REPO_ROOT/$OUT/prof1.err)
  echo "rocprof (1 stream) exit $?"
  python scripts/trace_by_grid.py $OUT/prof1 > $OUT/kernel_durations_by_grid_1stream.txt 2>&1; grep -E "final|istft|deconv2_stream|stft_forward_wave" $OUT/kernel_durations_by_grid_1stream.txt
fi
cat $OUT/mask_bins.txt 2>/dev/null | head -40
cat $OUT/f16_stats.txt 2>/dev/null
