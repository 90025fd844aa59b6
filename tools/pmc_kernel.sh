#!/bin/bash
# Developer tool (GPU box): hardware counters of ONE kernel family of the bench step, one rocprofv3 pass per counter group
# (PMC passes carry --kernel-trace only, as the pool requires).   usage: tools/pmc_kernel.sh <kernel-regex> <tag>
#   -> gpurun_out/<tag>_pmc_<group>.csv  (per-dispatch rows of the matching kernels, condensed by tools/pmc_post.py)
re=${1:-xattn_fused}; tag=${2:-pmc}
out=gpurun_out; mkdir -p $out; export TMPDIR=/tmp
groups=("FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU" "GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM")
i=0
for g in "${groups[@]}"; do
  d=$out/${tag}_pmc_$i
  rm -rf $d
  rocprofv3 --kernel-trace --pmc $g --kernel-include-regex "$re" --output-format csv -d $d -o run -- python bench.py --steps 2 --warmup 1 --headline-only > $out/${tag}_pmc_$i.log 2>&1
  i=$((i+1))
done
python tools/pmc_post.py $tag
