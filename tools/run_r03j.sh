cd $GRAFT_REPO_ROOT
CFD_DUAL_QKV=1 timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_sampler.py -m gpu -x -q -k "headline_shape_rows or (headline_shape_loop and ddpm5)" 2>&1 | tail -3
for v in 0 1 0 1; do
  CFD_DUAL_QKV=$v timeout 300 python bench.py --no-cpu-baseline --no-full-loop 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dual $v', round(d['value'],2), 'steps/s', {k:(v['ms'],v['launches']) for k,v in d['kernel_classes'].items()})"
done 2>&1 | grep "^dual"
