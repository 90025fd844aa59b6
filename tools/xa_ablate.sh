#!/bin/bash
# Timing ablations of xattn_fused_kernel (GPU box; garbage results): XA_ABLATE bit 1 = no fills, 2 = no MFMAs, 4 = no fragment reads, 8 = no softmax
cd "$(dirname "$0")/.."
for v in base 2 4 5 3; do
  unset CFD_LIB
  if [ $v != base ]; then export CFD_LIB=$PWD/tools/experiments/lib_xa$v.so; fi
  python tools/xa_ablate.py 2>/dev/null | tail -1 | sed "s/^/fused variant $v: /"
done
