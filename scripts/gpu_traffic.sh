#!/bin/bash
# HBM traffic of the kernels (separate --pmc passes, kernel-trace only), per MI355X_MICROARCH.md "HBM":
# FETCH_SIZE / WRITE_SIZE are in KiB-ish units of 1024 B?  -> we record raw values; bytes = value * 1024 for
# WRITE_SIZE; FETCH_SIZE is doubled on gfx950 for wide coalesced reads (the guide's correction).
set -u
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out; mkdir -p $OUT
cd /tmp
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 600 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/traffic_$c -o p -- \
     python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --sat-tiles 4096 > $OUT/traffic_$c.json 2> $OUT/traffic_$c.err
  echo "pmc $c exit $?"
done
# also a plain kernel-stats pass of the default bench for profiles/
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_final -o bench -- \
     python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline > $OUT/prof_final.json 2> $OUT/prof_final.err
echo "stats exit $?"
python - <<'PY'
import csv, glob, os, collections
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
res=collections.defaultdict(dict)
for c in ('FETCH_SIZE','WRITE_SIZE'):
    fs=glob.glob(out+'/traffic_%s/*counter_collection.csv'%c)
    rows=list(csv.DictReader(open(fs[0])))
    agg=collections.defaultdict(list)
    for r in rows:
        k=r['Kernel_Name']
        for key in ('final_kernel','deconv2','istft_wave','istft_fused','gemm_rows_splitk','gemm_rows_kernel','stft_forward'):
            if key in k: k=key; break
        agg[(k,int(r['Grid_Size']))].append(float(r['Counter_Value']))
    for k,v in agg.items(): res[k][c]=sum(v)/len(v)
for k in sorted(res): print(k, res[k])
PY
