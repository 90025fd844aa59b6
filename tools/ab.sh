#!/bin/bash
# Developer tool (GPU box): same-box A/B of two builds of the library on the bench workload, interleaved.
#   tools/ab.sh tools/experiments/lib_base.so convofusion_amd/libcfdenoise.so [rounds] [extra bench args]
A=$1; B=$2; R=${3:-2}; shift 3
for i in $(seq $R); do
  for lib in $A $B; do
    CFD_LIB=$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-loop "$@" 2>&1 | tail -1 | \
      python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],2), round(d['ms_per_step'],3), {k:v['ms'] for k,v in d['kernel_classes'].items() if v['ms']})"
  done
done
