cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03e
timeout 600 python tools/gemm_small_ab.py 1 6 19 20 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r03e/gemm_small.log
timeout 300 python bench.py --shape R --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/r03e/bench_R.json
python -c "
import json
d=json.load(open('gpurun_out/r03e/bench_R.json'))
print('R', round(d['value'],1), {k:(v['ms'],v['launches']) for k,v in d['kernel_classes'].items()})"
