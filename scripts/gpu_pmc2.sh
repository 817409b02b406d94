#!/bin/bash
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd /tmp
rocprofv3 -L 2>/dev/null | grep -o -E "SQ_[A-Z_0-9]*MFMA[A-Z_0-9]*|SQ_BUSY_CU_CYCLES|SQ_VALU_[A-Z_0-9]*|SQ_INST_CYCLES_[A-Z_0-9]*" | sort -u | tr '\n' ' ' > $OUT/pmc_list.txt; cat $OUT/pmc_list.txt; echo
run() { name=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/pmc2_$name -o p -- \
     python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --streams 1 --no-cpu-baseline --sat-tiles 4096 > $OUT/pmc2_$name.json 2> $OUT/pmc2_$name.err; echo "pmc2 $name exit $?"; }
run m GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS
python - <<'PY'
import csv, glob, os, collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
rows=list(csv.DictReader(open(glob.glob(out+'/pmc2_m/*counter_collection.csv')[0])))
agg=collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k=r['Kernel_Name']
    for key in ('final_kernel','deconv2_stream','deconv2','istft_wave','gemm_rows_splitk','gemm_rows_kernel','stft_forward_wave','stft_forward'):
        if key in k: k=key; break
    gs=int(r['Grid_Size'])
    agg[(k,gs)]['dur_us'].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
    agg[(k,gs)][r['Counter_Name']].append(float(r['Counter_Value']))
for (k,gs),d in sorted(agg.items()):
    m={c:sum(v)/len(v) for c,v in d.items()}
    if gs < 1000000: continue
    cu_cycles = m['GRBM_GUI_ACTIVE']/8
    print(k, gs, "dur %.1f us clk %.2f GHz | MFMA_BUSY %.3g (per CU-cycle*4simd: %.1f%%) INSTS_MFMA %.3g -> cycles/mfma %.1f | VALU active(quad) %.3g insts %.3g | wave_cycles(quad) %.3g" % (
        m['dur_us'], cu_cycles/(m['dur_us']*1e3), m['SQ_VALU_MFMA_BUSY_CYCLES'], 100*m['SQ_VALU_MFMA_BUSY_CYCLES']/(cu_cycles*256*4), m['SQ_INSTS_MFMA'], m['SQ_VALU_MFMA_BUSY_CYCLES']/max(m['SQ_INSTS_MFMA'],1), m['SQ_ACTIVE_INST_VALU'], m['SQ_INSTS_VALU'], m['SQ_WAVE_CYCLES']))
PY
