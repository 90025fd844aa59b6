# Developer tool (GPU box): what the LayerNorm fold costs its consumers -- builds with -DLNF_ABL=1 / 2 / 4 / 7 (gemm_sp.hpp) as tools/experiments/lib_lnf<v>.so
# (python -m convofusion_amd.build -DLNF_ABL=<v> -o tools/experiments/lib_lnf<v>.so) against the in-tree build on the R workload.  profiles/r06_ln_fold_ab.log
for r in 1 2; do
for v in 0 1 2 4 7; do
  if [ $v = 0 ]; then unset CFD_LIB; else export CFD_LIB=$PWD/tools/experiments/lib_lnf$v.so; fi
  python bench.py --steps 40 --warmup 3 --shape R --headline-only --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('LNF_ABL=$v', round(d['value'],2), 'steps/s', round(d['ms_per_step'],4), 'ms', {k:round(v['ms'],4) for k,v in d['kernel_classes'].items() if v['ms']})"
done
done
