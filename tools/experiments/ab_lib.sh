#!/bin/bash
# Developer tool (GPU box): tools/experiments/lib_base.so (another build) against the in-tree build, interleaved, at the headline shape (C2)
# and at the product shape (R).   tools/experiments/ab_lib.sh [rounds]
rounds=${1:-3}
for r in $(seq $rounds); do
  for which in base new; do
    if [ $which = base ]; then export CFD_LIB=$PWD/tools/experiments/lib_base.so; else unset CFD_LIB; fi
    for shape in C2 R; do
    extra=""; [ $shape = R ] && extra="--shape R"
    python bench.py --steps 30 --warmup 3 --headline-only $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which $shape', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms', {k:round(v['ms'],3) for k,v in d['kernel_classes'].items() if v['ms']})"
    done
  done
done
