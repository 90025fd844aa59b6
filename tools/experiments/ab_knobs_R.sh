#!/bin/bash
# Developer tool (GPU box): several settings of one environment knob at the product shape (32 utterances, bench.py --shape R; 8 through sample()), interleaved.
#   tools/experiments/ab_knobs_R.sh "CFD_QKV_CFG=1" "CFD_QKV_CFG=19" ...
for r in 1 2; do
for k in "" "$@"; do
  env $k python bench.py --steps 40 --warmup 3 --shape R --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$k] R', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms', {k:round(v['ms'],3) for k,v in d['kernel_classes'].items() if v['ms']})"
  env $k python tools/c1_time.py 8 2 2>/dev/null | tail -1 | sed "s/^/[$k] /"
done; done
