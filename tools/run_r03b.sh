set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03b
timeout 300 python tools/gemm_ab.py 1 30 > gpurun_out/r03b/gemm_ab.log 2>&1
cat gpurun_out/r03b/gemm_ab.log
timeout 600 python -m pytest tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -5
for cfg in 1 30 1 30; do
  CFD_BIG_CFG=$cfg timeout 300 python bench.py --no-cpu-baseline --no-full-loop 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('cfg $cfg', round(d['value'],2), 'steps/s', {k:v['ms'] for k,v in d['kernel_classes'].items()})"
done 2>&1 | tee gpurun_out/r03b/bench_ab.log
VARIANTS="7" timeout 900 tools/concurrency_variants.sh run 300 > gpurun_out/r03b/variants7.log 2>&1
cat gpurun_out/r03b/variants7.log
VARIANTS="6" timeout 1200 tools/concurrency_variants.sh run 1500 > gpurun_out/r03b/variants6.log 2>&1
cat gpurun_out/r03b/variants6.log
