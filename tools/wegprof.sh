#!/bin/bash
# GPU box: rocprofv3 kernel stats of WEG evaluations at the product shape (tools/weg_time.py: 53 full + 53 memory-side-reusing evaluations).
set -eu
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
TAG=${1:-wegprof}
mkdir -p gpurun_out
rm -rf gpurun_out/$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$TAG -o run -- python tools/weg_time.py > gpurun_out/$TAG.log 2>&1 || { rc=$?; echo "rocprofv3 run failed (rc $rc); log tail:" >&2; tail -20 gpurun_out/$TAG.log >&2; exit $rc; }
tail -2 gpurun_out/$TAG.log
python - "$TAG" <<'PY'
import csv,glob,sys,shutil
tag=sys.argv[1]
f=glob.glob(f'gpurun_out/{tag}/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
shutil.copy(f, f'gpurun_out/{tag}_kernel_stats.csv')
for r in rows[:30]:
    print(r['Name'][:100].ljust(100), r['Calls'].rjust(7), f"{float(r['AverageNs'])/1e3:8.1f} us", f"{float(r['TotalDurationNs'])/tot*100:5.1f}%")
print("total kernel time ms", tot/1e6)
PY
rm -rf gpurun_out/$TAG
