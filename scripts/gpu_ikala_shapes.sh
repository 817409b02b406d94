#!/bin/bash
# iKala conv2 / conv2^T (slabconv_ps.hip): parity of the tap-loop variants against the oracle, then the leg once per variant.
#   DCS_IKALA_VARIANTS="NAME=V;NAME2=V2 ..." (default: the old tap loop, DCS_SLABCONV_PS_FAST=0)
for v in "DCS_SLABCONV_PS_FAST=1" "DCS_SLABCONV_PS_FAST=0"; do
  echo "== parity $v"; env $v timeout 600 python -m pytest tests -m gpu -x -q -k "ikala_conv2_kernels or (ikala and full_size)" 2>&1 | tail -2
done
DCS_AB_LEGS=ikala DCS_AB_VARIANTS="${DCS_IKALA_VARIANTS:-DCS_SLABCONV_PS_FAST=0}" bash scripts/gpu_legs_ab.sh
