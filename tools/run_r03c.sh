set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03c
timeout 900 python -m pytest tests/test_gpu_forward.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -5
timeout 1200 python -m pytest tests/test_gpu_sampler.py -m gpu -x -q -k "headline or knobs or trajectory" 2>&1 | tail -8
for v in 0 1 0 1; do
  CFD_ROWLN=$v timeout 300 python bench.py --no-cpu-baseline --no-full-loop 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rowln $v', round(d['value'],2), 'steps/s', {k:(v['ms'],v['launches']) for k,v in d['kernel_classes'].items()})"
done 2>&1 | grep rowln | tee gpurun_out/r03c/bench_ab.log
