#!/bin/bash
# PMC passes over one bench leg (each its own rocprofv3 run with --kernel-trace only).
#   DCS_PMC_LEG=bach10_f16   DCS_PMC_ENV="NAME=V ..."   DCS_PMC_SETS="A,B,C;D,E"  (';' separates passes)
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
LEG=${DCS_PMC_LEG:-bach10_f16}
CMD="env ${DCS_PMC_ENV:-} python $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --streams 1 --no-cpu-baseline --no-host-fed --legs=$LEG --sat-tiles 0 --min-time 0.01"
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o "\b\(SQC\?_[A-Z_0-9]*\)\b" | sort -u > $OUT/pmc_avail_sq.txt
i=0
IFS=';' read -ra SETS <<< "${DCS_PMC_SETS:-SQ_WAVE_CYCLES,SQ_BUSY_CYCLES,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_VALU,SQ_INSTS_MFMA,SQ_IFETCH}"
for set in "${SETS[@]}"; do
  rm -rf $OUT/pmcleg_$i
  timeout 600 rocprofv3 --kernel-trace --pmc ${set//,/ } --output-format csv -d $OUT/pmcleg_$i -o p -- $CMD > $OUT/pmcleg_$i.json 2> $OUT/pmcleg_$i.err
  echo "pass $i ($set) exit $?"
  i=$((i+1))
done
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, os
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out"
rows = collections.defaultdict(lambda: collections.defaultdict(list))
for f in sorted(glob.glob(out + "/pmcleg_*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][-60:]
        rows[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/pmcleg_summary.txt", "w") as fh:
    for k, c in sorted(rows.items()):
        line = "%-62s n=%d " % (k, max(len(v) for v in c.values())) + "  ".join("%s=%.4g" % (n, sum(v) / len(v)) for n, v in sorted(c.items()))
        print(line); fh.write(line + "\n")
PY
