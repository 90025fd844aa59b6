cd $GRAFT_REPO_ROOT
for i in 1 2; do
for lib in convofusion_amd/libcfdenoise.so tools/experiments/lib_nt.so; do
  CFD_LIB=$PWD/$lib python tools/gemm_ab.py 1 2>&1 | grep "epi=0" | head -2 | sed "s|^|$lib |"
  CFD_LIB=$PWD/$lib python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-full-loop 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value'],2), round(d['ms_per_step'],3), {k:v['ms'] for k,v in d['kernel_classes'].items() if v['ms']})"
done; done
