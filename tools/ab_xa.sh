#!/bin/bash
# Developer tool (GPU box): the three forms of the fused cross-attention kernel (CFD_XA_PP = 0 / 1 / 2) on the bench workload, interleaved.
rounds=${1:-2}
for r in $(seq $rounds); do
  for f in 0 1 2; do
    CFD_XA_PP=$f python bench.py --steps 30 --warmup 3 --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('CFD_XA_PP=$f', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms', {k:round(v['ms'],3) for k,v in d['kernel_classes'].items() if v['ms']})"
  done
done
