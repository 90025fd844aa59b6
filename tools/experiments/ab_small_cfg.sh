#!/bin/bash
# R shape with several small-GEMM tile configs, interleaved
for r in 1 2; do
for k in "" "CFD_SMALL_CFG=27" "CFD_SMALL_CFG=25" "CFD_SMALL_CFG=23 CFD_SMALL_CFG_I=1024" "CFD_SMALL_CFG=24 CFD_SMALL_CFG_I=1024"; do
  env $k python bench.py --steps 40 --warmup 3 --shape R --headline-only 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('[$k] R', round(d['value'],2), 'steps/s', round(d['ms_per_step'],3), 'ms', {k:round(v['ms'],3) for k,v in d['kernel_classes'].items() if v['ms']})"
  env $k python tools/c1_time.py 8 2 2>/dev/null | tail -1 | sed "s/^/[$k] /"
done; done
