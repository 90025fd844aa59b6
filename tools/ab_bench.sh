#!/bin/bash
# Developer tool: A/B two builds of libcfdenoise.so on ONE box (devices differ by several per cent, so numbers from
# different gpurun calls do not compare).  usage: tools/ab_bench.sh <base.so> [rounds]   (new = the in-tree build)
base=$1; rounds=${2:-3}
for r in $(seq $rounds); do
  for which in base new; do
    if [ $which = base ]; then export CFD_LIB=$PWD/$base; else unset CFD_LIB; fi
    python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms', {k:round(v['ms'],3) for k,v in d['kernel_classes'].items()})"
  done
done
