#!/bin/bash
# Round 6, visit U: iKala transposed conv2 on column strips with half-block units (slabconv_ps_kernel<..., SPLIT>)
set -u
export TMPDIR=/tmp
OUT=gpurun_out; mkdir -p $OUT
timeout 1500 python -m pytest tests -m gpu -x -q --timeout=600 -p no:cacheprovider -k "ikala or slab or generic or variants or guard" > $OUT/r06_u_pytest.log 2>&1
echo "pytest exit $?"; tail -n 4 $OUT/r06_u_pytest.log | cut -c1-200
: > $OUT/r06_u_legs.txt
for rep in 1 2 3; do
for v in 1 0; do
DCS_SLABCONV_PS_SPLIT=$v timeout 600 python bench.py --steps 20 --warmup 5 --legs ikala --no-cpu-baseline --no-host-fed --no-cli --sat-tiles 0 > $OUT/r06_u.line 2> $OUT/r06_u.err || tail -n 5 $OUT/r06_u.err
python - "$v" <<'PY' | tee -a $OUT/r06_u_legs.txt
import json, sys
d = json.load(open("bench_detail.json"))
for k, L in (d.get("legs") or {}).items():
    if isinstance(L, dict) and "ms_per_clip" in L:
        print("split=%s %-8s %.4f ms/clip parity %s | %s" % (sys.argv[1], k, L["ms_per_clip"], (L.get("parity_check") or {}).get("ok"), " ".join("%s %.3f" % kv for kv in L["kernels_ms"].items())))
    elif isinstance(L, dict): print(k, L)
PY
done
done
