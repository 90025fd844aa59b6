#!/bin/bash
# GPU box: first checks of the row-tile path -- the small-shape parity tests, then the C1 wall time with and without it.
set -u
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
{
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_forward.py 2>&1 | tail -25
timeout 900 python -m pytest -x -q -m gpu tests/test_gpu_sampler.py -k "trajectory or inpaint or dedup or structured" 2>&1 | tail -15
for B in 1 2 4; do
  timeout 300 python tools/c1_time.py $B 2 2>&1 | tail -1
  CFD_ROWTILE=0 timeout 300 python tools/c1_time.py $B 2 2>&1 | tail -1
done
} > gpurun_out/rt_check.log 2>&1
tail -60 gpurun_out/rt_check.log
