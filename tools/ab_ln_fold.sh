#!/bin/bash
# Developer tool (GPU box): the algebraic LayerNorm fold of mid-size problems (CFD_LN_FOLD: default by shape, 0 off) at the product shape, interleaved on
# ONE box: the bench's R workload (32 utterances) and 1000-step runs through sample() for several batch sizes (tools/c1_time.py).
#   tools/ab_ln_fold.sh "8 16 32" [rounds]        (profiles/r06_ln_fold_ab.log)
Bs=${1:-"8 16 32"}; rounds=${2:-3}
for r in $(seq $rounds); do
  for fold in 0 -1; do
    export CFD_LN_FOLD=$fold
    for B in $Bs; do python tools/c1_time.py $B 2 2>/dev/null | tail -1 | sed "s/^/CFD_LN_FOLD=$fold /"; done
    python bench.py --steps 40 --warmup 3 --shape R --headline-only --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('CFD_LN_FOLD=$fold R', round(d['value'],2), 'steps/s', round(d['ms_per_step'],4), 'ms', {k:round(v['ms'],4) for k,v in d['kernel_classes'].items() if v['ms']})"
  done
done
