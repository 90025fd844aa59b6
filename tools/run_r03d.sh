set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03d
CFD_ROWLN_MIN_ROWS=1 timeout 900 python -m pytest tests/test_gpu_forward.py -m gpu -x -q 2>&1 | tail -3
for v in -1 16384 -1 16384; do
  CFD_ROWLN_MIN_ROWS=$v timeout 300 python bench.py --no-cpu-baseline --no-full-loop 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('rowln $v', round(d['value'],2), 'steps/s', {k:(v['ms'],v['launches']) for k,v in d['kernel_classes'].items()})"
done 2>&1 | grep "^rowln" | tee gpurun_out/r03d/bench_ab.log
