#!/bin/bash
# Developer tool (GPU box): one environment knob at the product shape (R: 32 utterances; and 8 / 16 utterances through sample()), interleaved.
#   tools/experiments/ab_knob_R.sh CFD_QKV_FUSED=0 [rounds]
knob=$1; rounds=${2:-2}
for r in $(seq $rounds); do
for k in "" "$knob"; do
  env $k python bench.py --steps 40 --warmup 3 --shape R --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$k] R', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms', {k:round(v['ms'],3) for k,v in d['kernel_classes'].items() if v['ms']})"
  for B in 8 16; do env $k python tools/c1_time.py $B 2 2>/dev/null | tail -1 | sed "s/^/[$k] /"; done
done; done
