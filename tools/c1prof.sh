#!/bin/bash
# GPU box: rocprofv3 kernel stats of one utterance at the product shape (C1), 1000-step DDPM.
set -eu
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
mkdir -p gpurun_out
rm -rf gpurun_out/c1prof
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/c1prof -o run -- python tools/c1_time.py 1 1 > gpurun_out/c1prof.log 2>&1
tail -3 gpurun_out/c1prof.log
python - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/c1prof/**/*kernel_stats.csv',recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r['TotalDurationNs']) for r in rows)
for r in rows[:22]:
    print(r['Name'][:95].ljust(95), r['Calls'].rjust(7), f"{float(r['AverageNs'])/1e3:8.1f} us", f"{float(r['TotalDurationNs'])/tot*100:5.1f}%")
print("total kernel time ms", tot/1e6)
PY
