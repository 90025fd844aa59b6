#!/bin/bash
# Two-queue investigation.  HISTORICAL: the CFD_EXP variants and the CFD_EAGER_STEPS knob it drives left the product
# sources in round 4 (tools/experiments/r03_variants/removed_from_product.patch restores them on the round-3 tree).  BUILD (container, no GPU):  tools/concurrency_variants.sh build
#   -> tools/experiments/lib_exp<k>.so, k = CFD_EXP variant (gemm_sp.hpp / rows.hpp):
#      1 s_waitcnt vmcnt(0) at the end of every gemm_sp_kernel wave      2 (1) + eps stored with sc1 (write-through)
#      3 (1) + eps stored sc0 sc1                                         4 agent-scope release fence at the end of every gemm wave
#      5 agent-scope acquire at the head of cfg_step_kernel               6 ~10 us delay at the head of cfg_step_kernel
#      7 cfg_step_kernel checks (device printf) that every workgroup of the final projection had finished when it started
# RUN (GPU box):  tools/concurrency_variants.sh run [REPS]   -- the soak with eagerly enqueued iterations (CFD_EAGER_STEPS=1:
#   46 of 600 repetitions differed in round 2) for the product build and every variant.
set -u
cd "$(dirname "$0")/.."
if [ "${1:-}" = build ]; then
  for k in ${VARIANTS:-1 2 3 4 5 6 7}; do
    python -m convofusion_amd.build -DCFD_EXP=$k -o tools/experiments/lib_exp$k.so
  done
  wait
  ls -la tools/experiments/lib_exp*.so
else
  REPS=${2:-300}
  for v in ${VARIANTS:-base 1 2 3 4 5 6 7}; do
    if [ $v = base ]; then unset CFD_LIB; else export CFD_LIB=$PWD/tools/experiments/lib_exp$v.so; fi
    echo "== variant $v"
    CFD_EAGER_STEPS=${EAGER-1} REPS=$REPS timeout 900 python tools/concurrency_soak.py 32 8 2>&1 | grep -v "^rep" | tail -8
  done
fi
