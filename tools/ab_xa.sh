#!/bin/bash
# Developer tool (GPU box): the operand policies of the fused cross-attention kernel (CFD_XA_OPERANDS = 0 pairs / 15 single-fp16 operands
# against the long memories; developer builds with -DXA_ALL_OPF=1 also take 3 / 7 / 11; xattn_fused.hpp OPF) on the bench workload, interleaved on ONE box.  usage: tools/ab_xa.sh [rounds] [modes...]
rounds=${1:-2}
shift
modes=${@:-0 15}
for r in $(seq $rounds); do
  for f in $modes; do
    CFD_XA_OPERANDS=$f python bench.py --steps 30 --warmup 3 --headline-only --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('CFD_XA_OPERANDS=$f', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms', {k:round(v['ms'],3) for k,v in d['kernel_classes'].items() if v['ms']})"
  done
done
