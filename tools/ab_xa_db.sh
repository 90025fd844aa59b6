#!/bin/bash
# Developer tool (GPU box, developer build: python -m convofusion_amd.build -DXA_ALL_OPF=1): operand policy 15 of the fused cross-attention on the
# three-barrier step (CFD_XA_DB=0) against the shipped double-buffered step (CFD_XA_DB=1; xattn_fused.hpp XA_DBUF), interleaved on ONE box at the
# product shape (R) and the headline shape (C2).  usage: tools/ab_xa_db.sh [rounds]      (profiles/r06_xa_dbuf_ab.log)
for r in $(seq ${1:-3}); do
 for db in 0 1; do
  for shape in R C2; do
   CFD_XA_DB=$db timeout 300 python bench.py --shape $shape --steps 40 --warmup 5 --headline-only --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('CFD_XA_DB=$db $shape', round(d['value'],2), 'steps/s', round(d['ms_per_step'],4), 'ms', {k:round(v['ms'],4) for k,v in d['kernel_classes'].items() if v['ms']})"
  done
 done
done
