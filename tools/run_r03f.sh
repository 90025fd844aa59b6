cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03f
export CFD_XA_ROLE=1
timeout 600 python -m pytest tests/test_gpu_forward.py -m gpu -x -q 2>&1 | tail -5
timeout 900 python -m pytest tests/test_gpu_sampler.py -m gpu -x -q -k "headline_shape_loop_row_matches_reference and ddpm5 or several_long or trajectory" 2>&1 | tail -5
for v in 0 1 0 1; do
  CFD_XA_ROLE=$v timeout 300 python bench.py --no-cpu-baseline --no-full-loop 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('role $v', round(d['value'],2), 'steps/s', {k:(v['ms'],v['launches']) for k,v in d['kernel_classes'].items()})"
done 2>&1 | grep "^role" | tee gpurun_out/r03f/bench_ab2.log
