#!/bin/bash
# two builds interleaved on one box: base = tools/experiments/lib_base.so, new = the in-tree build; shapes C2 and R
for r in 1 2 3; do
  for which in base new; do
    if [ $which = base ]; then export CFD_LIB=$PWD/tools/experiments/lib_base.so; else unset CFD_LIB; fi
    for shape in C2 R; do
      python bench.py --steps 30 --warmup 3 --shape $shape --no-cpu-baseline --no-secondary --no-full-loop 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which $shape', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms')"
    done
  done
done
