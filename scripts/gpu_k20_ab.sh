#!/bin/bash
# Round 4: A/B at the DRIVER's launch shape (bench.py --steps 20: one launch group of 20 x 32 tiles on one stream), then a
# rocprofv3 kernel trace of that same command (the trace at grid 552 960 the round-3 review asked for).
#   DCS_AB_VARIANTS="default NAME=VAL DCS_LIB=deepconvsep_amd/_exp_x.so ..."   DCS_K20_TRACE=0 skips the trace
#   DCS_AB_K="expr" runs `pytest -m gpu -k expr` first
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
if [ -n "${DCS_AB_K:-}" ]; then
  timeout 1200 python -m pytest tests -m gpu -q -x --timeout=300 -p no:cacheprovider -k "$DCS_AB_K" > $OUT/k20_pytest.log 2>&1; echo "pytest exit $?"; tail -n 8 $OUT/k20_pytest.log
fi
: > $OUT/k20_ab.txt
for v in ${DCS_AB_VARIANTS:-default}; do
  envs=""; [ "$v" != "default" ] && envs="${v//+/ }"
  case "$envs" in DCS_LIB=*) envs="DCS_LIB=$PWD/${envs#DCS_LIB=}";; esac
  vn=${v//\//_}
  for rep in $(seq 1 ${DCS_K20_REPS:-2}); do
    env $envs timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-fed --no-cli --legs= --sat-tiles 0 \
        ${DCS_K20_ARGS:-} > $OUT/k20_$vn.line 2> $OUT/k20_$vn.err || { echo "== $v FAILED"; tail -n 3 $OUT/k20_$vn.err; }
    cp bench_detail.json $OUT/k20_$vn.json 2>/dev/null   # the full result (the stdout line is the compact headline)
    python - "$v" "$OUT/k20_$vn.json" <<'PY' | tee -a $OUT/k20_ab.txt
import json, sys
try:
    d = json.loads(open(sys.argv[2]).read().strip().splitlines()[-1])
except Exception as e:
    print("%-40s unreadable: %s" % (sys.argv[1], e)); sys.exit(0)
g = d["launch_group"]["kernels_ms"]
pc = d.get("parity_check") or {}
print("%-40s %.5f ms/step  %.2f M frames/s  whole %.3f  final %.1f us frac %.3f  parity %s %.2g | group us: %s | single %.4f"
      % (sys.argv[1], d["ms_per_step"], d["value"] / 1e6, d["whole_path_frac_of_f32_peak"], 1e3 * d["roofline"]["avg_kernel_ms"],
         d["roofline"]["frac"], pc.get("ok"), pc.get("max_abs_pcm_err", float("nan")),
         " ".join("%s %.1f" % (t, 1e3 * g[t]) for t in g), d["single_stream"]["ms_per_step"]))
PY
  done
done
if [ "${DCS_K20_TRACE:-1}" = "1" ]; then
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$OUT/prof_k20 -o bench -- \
      python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-host-fed --no-cli --legs= --sat-tiles 0 --no-parity-check \
      > $GRAFT_REPO_ROOT/$OUT/prof_k20_bench.line 2> $GRAFT_REPO_ROOT/$OUT/prof_k20.err)
  cp bench_detail.json $OUT/prof_k20_bench.json 2>/dev/null
  echo "rocprof exit $?"
  python scripts/trace_by_grid.py $OUT/prof_k20 > $OUT/kernel_durations_by_grid_k20.txt 2>&1
  grep -E "final|istft|deconv2|stft_forward|gemm" $OUT/kernel_durations_by_grid_k20.txt
  for f in $(find $OUT/prof_k20 -name "*kernel_stats*.csv" | head -1); do cp $f $OUT/k20_kernel_stats.csv; done
  rm -rf $OUT/prof_k20
fi
