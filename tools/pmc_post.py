"""Condense the per-dispatch counter CSVs of tools/pmc_kernel.sh into one table: mean per launch of every counter, per kernel.
usage: python tools/pmc_post.py <tag>   (reads gpurun_out/<tag>_pmc_*/, writes gpurun_out/<tag>_pmc_summary.csv)"""
import csv
import glob
import os
import sys
from collections import defaultdict

tag = sys.argv[1]
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
acc = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
dur = defaultdict(lambda: [0.0, 0])
for f in glob.glob(os.path.join(OUT, f"{tag}_pmc_*", "**", "*counter_collection.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        a = acc[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"])
        a[1] += 1
        if "Start_Timestamp" in r and "End_Timestamp" in r:
            d = dur[k]
            d[0] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
            d[1] += 1
names = sorted({c for k in acc for c in acc[k]})
with open(os.path.join(OUT, f"{tag}_pmc_summary.csv"), "w", newline="") as fo:
    w = csv.writer(fo)
    w.writerow(["kernel", "launches", "avg_us_under_pmc"] + names)
    for k in acc:
        n = max(v[1] for v in acc[k].values())
        w.writerow([k, n, round(dur[k][0] / max(dur[k][1], 1) / 1e3, 1)] + [round(acc[k][c][0] / max(acc[k][c][1], 1), 1) if c in acc[k] else "" for c in names])
print(open(os.path.join(OUT, f"{tag}_pmc_summary.csv")).read())
