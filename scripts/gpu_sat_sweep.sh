#!/bin/bash
# per-kernel milliseconds of one clip of T tiles (bench.py's saturating leg) for a list of sizes and environment variants:
#   DCS_SWEEP_TILES="320 640 ..."  DCS_SWEEP_VARIANTS="default NAME=VAL ..."
for t in ${DCS_SWEEP_TILES:-320 640 1024 2048 4096}; do
  for v in ${DCS_SWEEP_VARIANTS:-default}; do
    e=""; [ "$v" != default ] && e="$v"
    env $e python bench.py --steps 32 --warmup 8 --min-time 0.05 --sat-tiles $t --legs= --no-cpu-baseline --no-host-fed --no-cli --no-parity-check 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); s=d['saturating']
print('tiles %5d %-24s %.4f ms' % ($t, '$v', s['ms_per_step']), {k: round(x*1e3,1) for k,x in s['kernels_ms'].items()})"
  done
done
