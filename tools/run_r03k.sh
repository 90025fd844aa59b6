cd $GRAFT_REPO_ROOT
for i in 1 2 3 4; do for v in 0 1; do
  CFD_DUAL_QKV=$v timeout 300 python bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-full-loop 2> /dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('dual $v', round(d['ms_per_step'],4), round(d['value'],2))"
done; done 2>&1 | grep "^dual"
