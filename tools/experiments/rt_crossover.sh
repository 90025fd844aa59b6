#!/bin/bash
# Developer tool (GPU box): the row-tile path against the tile kernels at the product shape for several utterance counts (1000-step runs through sample()).
#   tools/experiments/rt_crossover.sh "4 5 6 7"
for B in ${1:-"4 5 6 7"}; do
  CFD_ROWTILE_MAX_ROWS=100000 python tools/c1_time.py $B 2 2>/dev/null | tail -1 | sed "s/^/rowtile  /"
  CFD_ROWTILE=0 python tools/c1_time.py $B 2 2>/dev/null | tail -1 | sed "s/^/tile     /"
done
