cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/qtrace_bench -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 400 --warmup 40 --no-cpu-baseline --sat-tiles 0 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/qtrace_exp -o t -- python $GRAFT_REPO_ROOT/scripts/gpu_host_exp.py > /dev/null 2>&1
python - <<'PY'
import csv, collections, os
for name in ('qtrace_bench','qtrace_exp'):
    rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/%s/t_kernel_trace.csv'%name)))
    rows.sort(key=lambda r:int(r['Start_Timestamp']))
    # last 4000 kernels ~ multi-stream phase? use stream/queue histogram over whole run
    q=collections.Counter((r['Queue_Id'],r['Stream_Id']) for r in rows)
    print(name, 'kernels', len(rows), 'distinct (queue,stream):', len(q))
    byq=collections.Counter(r['Queue_Id'] for r in rows); print('  per queue', dict(byq))
    bys=collections.defaultdict(set)
    for (qq,ss),c in q.items(): bys[qq].add(ss)
    print('  streams per queue', {k:len(v) for k,v in bys.items()})
PY
