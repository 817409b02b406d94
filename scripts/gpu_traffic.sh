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
     python $GRAFT_REPO_ROOT/bench.py --steps 32 --warmup 16 --no-cpu-baseline --sat-tiles 4096 > $OUT/traffic_$c.json 2> $OUT/traffic_$c.err
  echo "pmc $c exit $?"
done
python - <<'PY'
import csv, glob, os, collections, json
out=os.environ['GRAFT_REPO_ROOT']+'/gpurun_out'
res=collections.defaultdict(dict)
for c in ('FETCH_SIZE','WRITE_SIZE'):
    fs=glob.glob(out+'/traffic_%s/**/*counter_collection.csv'%c, recursive=True)
    rows=list(csv.DictReader(open(fs[0])))
    agg=collections.defaultdict(list)
    for r in rows:
        k=r['Kernel_Name']
        for key in ('final_kernel','deconv2_stream','deconv2','istft_wave','istft_fused','gemm_rows_splitk','gemm_rows_kernel','stft_forward_wave','stft_forward','copyBuffer'):
            if key in k: k=key; break
        agg['%s@grid_threads=%d' % (k,int(r['Grid_Size']))].append(float(r['Counter_Value']))
    for k,v in agg.items(): res[k][c+'_KiB']=round(sum(v)/len(v),2)
doc={'all':{k:res[k] for k in sorted(res)}}
fin=sorted((int(k.split('=')[1]),k) for k in res if k.startswith('final_kernel@'))
if fin:
    doc['final_kernel_32_tiles']=res[fin[0][1]]; doc['final_kernel_4096_tiles']=res[fin[-1][1]]
    if len(fin) >= 3:   # the launch group of the headline run: clips_per_launch x 32 tiles
        cfg=json.loads(open(out+'/traffic_FETCH_SIZE.json').read().strip().splitlines()[-1])['config']
        doc['final_kernel_%dx%d_tiles' % (cfg['clips_per_launch'], cfg['tiles_per_gpu_per_step'])]=res[fin[1][1]]
doc['note']=('rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes (scripts/gpu_traffic.sh), '
  'bench.py --steps 32 --sat-tiles 4096, MI355X; per launch, KiB (averaged over launches of the same kernel and grid). '
  'MI355X_MICROARCH.md: on gfx950 FETCH_SIZE counts half the bytes of 16-byte-per-lane coalesced reads; bench.py reports '
  'traffic = 2*FETCH + WRITE as an upper bound.')
json.dump(doc, open(out+'/traffic.json','w'), indent=1)
for k in sorted(res): print(k, res[k])
PY
