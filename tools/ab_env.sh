#!/bin/bash
# Developer tool: A/B one environment knob of libcfdenoise on ONE box, interleaved.  usage: tools/ab_env.sh CFD_XA_PP=0 [rounds] [extra bench args]
knob=$1; rounds=${2:-2}; shift 2
for r in $(seq $rounds); do
  for which in default "$knob"; do
    if [ "$which" = default ]; then pre=""; else pre="$knob"; fi
    env $pre python bench.py --steps 30 --warmup 3 --headline-only "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$which', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms', {k:round(v['ms'],3) for k,v in d['kernel_classes'].items()})"
  done
done
