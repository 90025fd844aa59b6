#!/bin/bash
# GPU box: rocprofv3 kernel stats of B utterances (default 1) at the product shape (C1), 1000-step DDPM.
set -eu
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
B=${1:-1}
TAG=${2:-c1prof}
mkdir -p gpurun_out
rm -rf gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -o run -- python tools/c1_time.py $B 1 > gpurun_out/$TAG.log 2>&1 || { rc=$?; echo "rocprofv3 run failed (rc $rc); log tail:" >&2; tail -20 gpurun_out/$TAG.log >&2; exit $rc; }
tail -3 gpurun_out/$TAG.log
python - "$TAG" <<'PY'
import csv,glob,sys
tag=sys.argv[1]
f=glob.glob(f'gpurun_out/{tag}/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
import shutil; shutil.copy(f, f'gpurun_out/{tag}_kernel_stats.csv')
for r in rows[:26]:
    print(r['Name'][:95].ljust(95), r['Calls'].rjust(7), f"{float(r['AverageNs'])/1e3:8.1f} us", f"{float(r['TotalDurationNs'])/tot*100:5.1f}%")
print("total kernel time ms", tot/1e6)
PY
rm -rf gpurun_out/$TAG
