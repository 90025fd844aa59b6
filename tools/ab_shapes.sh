#!/bin/bash
# Developer tool (GPU box): tools/experiments/lib_base.so (a build of an earlier commit) against the in-tree build at the product shape for
# several batch sizes (1000-step runs through sample(), tools/c1_time.py) and on the bench workloads (headline C2, product shape R), interleaved.
#   tools/ab_shapes.sh "8 16 32" [rounds]
Bs=${1:-"8 16 32"}; rounds=${2:-2}
for r in $(seq $rounds); do
  for which in base new; do
    if [ $which = base ]; then export CFD_LIB=$PWD/tools/experiments/lib_base.so; else unset CFD_LIB; fi
    for B in $Bs; do python tools/c1_time.py $B 2 2>/dev/null | tail -1 | sed "s/^/$which /"; done
    python bench.py --steps 40 --warmup 3 --shape R --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which R', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms', {k:round(v['ms'],3) for k,v in d['kernel_classes'].items() if v['ms']})"
  done
done
