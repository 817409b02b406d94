#!/bin/bash
# PMC pass (own run, kernel-trace only): effective clock and wait breakdown per kernel.
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd /tmp
run() { # name, counters...
  name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc_$name -o p -- \
     python $GRAFT_REPO_ROOT/bench.py --steps ${STEPS:-20} --warmup 3 --streams ${STREAMS:-2} --no-cpu-baseline --sat-tiles ${SAT:-1024} > $OUT/pmc_$name.json 2> $OUT/pmc_$name.err
  echo "pmc $name exit $?"
}
run a GRBM_GUI_ACTIVE SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU
run b GRBM_GUI_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM SQ_INSTS_SALU SQ_INSTS_MFMA
ls $OUT/pmc_a | head; 
python - <<'PY'
import csv, glob, os, collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
for tag in 'ab':
    fs=glob.glob(out+'/pmc_%s/*counter_collection.csv'%tag)
    if not fs: print('no csv', tag); continue
    rows=list(csv.DictReader(open(fs[0])))
    print(tag, len(rows), rows[0].keys())
    agg=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        k=r['Kernel_Name']
        for key in ('final_bf16x3','g_split','final_kernel','deconv2_stream','deconv2','istft_wave','gemm_rows_splitk','gemm_rows_kernel','stft_forward_wave','stft_forward','conv1','colconv'):
            if key in k: k=key; break
        gs=r.get('Grid_Size','?')
        if int(gs) < int(os.environ.get('MIN_GRID','0')): continue
        agg[(k,gs)][r['Counter_Name']].append(float(r['Counter_Value']))
    for (k,gs),d in sorted(agg.items()):
        print(k[:24], gs, {c: round(sum(v)/len(v),1) for c,v in d.items()}, 'n', len(next(iter(d.values()))))
PY
