"""Condense the raw outputs of tools/profile_round.sh into profiles/ (tracked).
  --aggregate-only (on the GPU box): per-(kernel, grid) means of FETCH_SIZE / WRITE_SIZE -> gpurun_out/<tag>_hbm_traffic.csv;
      per-kernel MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) -> gpurun_out/<tag>_mfma_busy.csv
  default (build container): copy the bench lines, the rocprofv3 kernel-stats summary and the traffic table into
  profiles/<tag>_*, and write profiles/hbm_traffic_gemm.json (the figure bench.py reports as roofline.traffic).
FETCH_SIZE is doubled: on gfx950 it counts 128-byte requests as 64 bytes (MI355X_MICROARCH.md, HBM)."""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

tag = sys.argv[1]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def aggregate():
    acc = defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0]})
    for name in ("fetch", "write"):
        for f in glob.glob(os.path.join(OUT, f"{tag}_pmc_{name}", "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                a = acc[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"])
                a[1] += 1
    rows = []
    for (k, g), v in acc.items():
        fs, fn = v["FETCH_SIZE"]
        ws, wn = v["WRITE_SIZE"]
        if fn == 0 and wn == 0:
            continue
        rows.append((k, g, max(fn, wn), fs / max(fn, 1), 2 * fs / max(fn, 1) / 1024, ws / max(wn, 1) / 1024))
    rows.sort(key=lambda r: -(r[4] + r[5]) * r[2])
    with open(os.path.join(OUT, f"{tag}_hbm_traffic.csv"), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "grid_threads", "launches_in_trace", "FETCH_SIZE_KB_raw_per_launch", "read_MB_per_launch_(2xFETCH)",
                    "write_MB_per_launch_(WRITE_SIZE)"])
        for r in rows:
            w.writerow([r[0], r[1], r[2], round(r[3]), round(r[4], 1), round(r[5], 1)])


def aggregate_mfma():
    """MFMA-pipe utilisation per kernel: busy cycles summed over the chip's 1024 SIMDs / (kernel cycles x 1024).  GRBM_GUI_ACTIVE is
    reported as the sum over the 8 XCDs (MI355X_MICROARCH.md, DVFS give-back)."""
    acc = defaultdict(lambda: {"SQ_VALU_MFMA_BUSY_CYCLES": 0.0, "GRBM_GUI_ACTIVE": 0.0, "n": 0})
    for f in glob.glob(os.path.join(OUT, f"{tag}_pmc_mfma", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = acc[r["Kernel_Name"]]
            a[r["Counter_Name"]] += float(r["Counter_Value"])
            if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
                a["n"] += 1
    rows = []
    for k, v in acc.items():
        if v["GRBM_GUI_ACTIVE"] <= 0:
            continue
        cyc = v["GRBM_GUI_ACTIVE"] / 8.0
        rows.append((k, v["n"], cyc / max(v["n"], 1), v["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), cyc))
    rows.sort(key=lambda r: -r[4])
    with open(os.path.join(OUT, f"{tag}_mfma_busy.csv"), "w", newline="") as fo:
        w = csv.writer(fo)
        w.writerow(["kernel", "launches_in_trace", "mean_kernel_cycles", "mfma_busy_fraction", "share_of_traced_gpu_cycles"])
        tot = sum(r[4] for r in rows) or 1.0
        for r in rows:
            w.writerow([r[0], r[1], round(r[2]), round(r[3], 4), round(r[4] / tot, 4)])


def publish():
    os.makedirs(PROF, exist_ok=True)
    for src, dst in ((f"{tag}_bench_c2.json", f"{tag}_bench_c2.json"), (f"{tag}_bench_realshape.json", f"{tag}_bench_realshape.json"),
                     (f"{tag}_rocprof_stdout.log", f"{tag}_bench_c2_rocprof_stdout.log"), (f"{tag}_hbm_traffic.csv", f"{tag}_bench_c2_hbm_traffic.csv"),
                     (f"{tag}_mfma_busy.csv", f"{tag}_mfma_busy.csv")):
        p = os.path.join(OUT, src)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(PROF, dst))
    for src, dst in ((f"{tag}_c1_kernel_stats.csv", f"{tag}_c1_single_utterance_kernel_stats.csv"), (f"{tag}_weg_kernel_stats.csv", f"{tag}_weg_eval_kernel_stats.csv"),
                     (f"{tag}_c1.txt", f"{tag}_c1_single_utterance_summary.txt"), (f"{tag}_weg.txt", f"{tag}_weg_eval_summary.txt")):
        p = os.path.join(OUT, src)
        if os.path.exists(p):
            shutil.copy(p, os.path.join(PROF, dst))
    b = os.path.join(OUT, f"{tag}_bench_c2.json")
    if os.path.exists(b):     # the other BASELINE configurations travel inside the bench line since round 4
        try:
            d = json.loads(open(b).read().strip().splitlines()[-1])
            oc = dict(d.get("other_configs") or {})
            oc["c1_single_utterance"] = d.get("c1_single_utterance")
            json.dump(oc, open(os.path.join(PROF, f"{tag}_other_configs.json"), "w"), indent=1)
        except Exception as e:
            print("other_configs not extracted:", e)
    stats = glob.glob(os.path.join(OUT, f"{tag}_stats", "**", "*kernel_stats.csv"), recursive=True)
    if stats:
        shutil.copy(stats[0], os.path.join(PROF, f"{tag}_bench_c2_kernel_stats.csv"))
    t = os.path.join(PROF, f"{tag}_bench_c2_hbm_traffic.csv")
    if os.path.exists(t):
        tot, n = 0.0, 0
        for r in csv.DictReader(open(t)):
            # the products launched inside an iteration (the memory-side projections and the timestep tables run once per run: EpiMemK / EpiMemV / EpiF32)
            if "gemm_sp_kernel" in r["kernel"] and not any(e in r["kernel"] for e in ("EpiMemK", "EpiMemV", "EpiF32")):
                k = int(r["launches_in_trace"])
                tot += (float(r["read_MB_per_launch_(2xFETCH)"]) + float(r["write_MB_per_launch_(WRITE_SIZE)"])) * 1e6 * k
                n += k
        xt, xn = 0.0, 0
        for r in csv.DictReader(open(t)):    # the full-batch launches of the fused cross-attention kernel (the largest grid in the trace)
            if "xattn_fused_kernel" in r["kernel"] or r["kernel"].startswith("xattn_role_kernel"):
                xt, xn = max((xt, xn), ((float(r["read_MB_per_launch_(2xFETCH)"]) + float(r["write_MB_per_launch_(WRITE_SIZE)"])) * 1e6, int(r["launches_in_trace"])))
        json.dump({"bytes_per_launch_mean": tot / max(n, 1), "launches_in_trace": n,
                   "xattn_bytes_per_launch": xt or None, "xattn_launches_in_trace": xn,
                   "source": f"profiles/{tag}_bench_c2_hbm_traffic.csv (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over "
                             "`bench.py --steps 3 --warmup 1`; FETCH doubled per MI355X_MICROARCH.md HBM note for gfx950)"},
                  open(os.path.join(PROF, "hbm_traffic_gemm.json"), "w"), indent=1)
    print("published:", sorted(os.listdir(PROF)))


def roofline():
    """profiles/<tag>_roofline.json: the judge's per-kernel table as a file.  Per (symbol, grid) of the rocprofv3 kernel trace of the bench
    command: launches per loop iteration, mean duration, ms per iteration, MFMA-busy (per symbol), HBM read / written per launch (per symbol and
    grid; FETCH doubled) -- and per kernel class the algorithmic FLOPs of the bench line over the class's traced time (frac of the 2.5 PFLOP/s
    dense f16 peak; x3 issued), plus the compulsory bytes of the three large product groups.  Iterations are counted by cfg_step_kernel (one per
    loop iteration, eager warm-up and replays alike); the eager profiled forward adds one more launch of every forward kernel."""
    tr = glob.glob(os.path.join(OUT, f"{tag}_stats", "**", "*kernel_trace.csv"), recursive=True)
    b = os.path.join(OUT, f"{tag}_bench_c2.json")
    if not tr or not os.path.exists(b):
        print("roofline: no kernel trace / bench line for", tag)
        return
    d = json.loads(open(b).read().strip().splitlines()[-1])
    acc = defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(tr[0])):
        k = (r["Kernel_Name"], int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"]))
        acc[k][0] += 1
        acc[k][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3
    def bare(name):       # (since round 6 every kernel is a template instance: rocprofv3 prints "void name<0>(...)")
        return name[5:] if name.startswith("void ") else name
    n_iter = sum(v[0] for k, v in acc.items() if bare(k[0]).startswith("cfg_step_kernel"))
    n_fwd = n_iter + 1
    busy, traffic = {}, {}
    p = os.path.join(OUT, f"{tag}_mfma_busy.csv")
    if os.path.exists(p):
        busy = {r["kernel"]: float(r["mfma_busy_fraction"]) for r in csv.DictReader(open(p))}
    p = os.path.join(OUT, f"{tag}_hbm_traffic.csv")
    if os.path.exists(p):
        traffic = {(r["kernel"], int(r["grid_threads"])): (float(r["read_MB_per_launch_(2xFETCH)"]), float(r["write_MB_per_launch_(WRITE_SIZE)"])) for r in csv.DictReader(open(p))}

    def cls(name):
        name = bare(name)
        if "gemm_sp_kernel" in name:
            return "gemm_mem" if ("EpiMemK" in name or "EpiMemV" in name) else "gemm_token"
        if "xattn_fused_kernel" in name:
            return "xattn"
        if "self_attn_fused_kernel" in name:
            return "gemm_attn"
        if "ln_rows_kernel" in name or name.startswith(("mem_scale", "replicate_rows", "begin_step", "cfg_step", "rt_step_rows", "att_fixup")):
            return "rows"
        return None
    rows, cms = [], defaultdict(float)
    for (name, grid), (n, us) in acc.items():
        per_step = n / n_fwd
        if per_step < 0.9 or cls(name) is None:      # once-per-run work (set-up, folding, tables): not part of an iteration
            continue
        ms_step = us / n_fwd * 1e-3
        cms[cls(name)] += ms_step
        rd, wr = traffic.get((name, grid), (None, None))
        rows.append({"kernel": name[:110], "grid_threads": grid, "class": cls(name), "launches_per_step": round(per_step, 2), "avg_us": round(us / n, 2),
                     "ms_per_step": round(ms_step, 4), "mfma_busy": busy.get(name), "read_MiB_per_launch": None if rd is None else round(rd * 1e6 / 2**20, 1),
                     "write_MiB_per_launch": None if wr is None else round(wr * 1e6 / 2**20, 1)})
    rows.sort(key=lambda r: -r["ms_per_step"])
    # compulsory HBM bytes of the three large product groups at the headline shape (split-pair = 4 B per element, float32 residual stream)
    M = 7 * 32 * 196
    MiB = 2.0**20
    comp = {"EpiResid N=512 (Wo, TB1, TB2: K=512; FFN2: K=1024)": {"read_MiB": "%.1f (K=512) / %.1f (K=1024)" % ((M * 512 * 4 * 2 + 512 * 512 * 4) / MiB, (M * 1024 * 4 + M * 512 * 4 + 512 * 1024 * 4) / MiB),
                                                                   "write_MiB": round(M * 512 * 4 / MiB, 1)},
            "EpiSplit N=1024 (q|k, FFN1)": {"read_MiB": round((M * 512 * 4 + 1024 * 512 * 4) / MiB, 1), "write_MiB": round(M * 1024 * 4 / MiB, 1)},
            "EpiSplit N=512 (v^T)": {"read_MiB": round((M * 512 * 4 + 512 * 512 * 4) / MiB, 1), "write_MiB": round(M * 512 * 4 / MiB, 1)},
            "xattn_fused_kernel": {"read_MiB": "x rows twice (LayerNorm2 + flush) %.1f + K / V^T tiles once per XCD that streams them" % (2 * M * 512 * 4 / MiB), "write_MiB": round(M * 512 * 4 / MiB, 1)},
            "self_attn_fused_kernel": {"read_MiB": round((M * 1024 * 4 + M * 512 * 4) / MiB, 1), "write_MiB": round(M * 512 * 4 / MiB, 1)},
            "ln_rows_kernel": {"read_MiB": round(M * 512 * 4 / MiB, 1), "write_MiB": round(M * 512 * 4 / MiB, 1)}}
    kc = d.get("kernel_classes", {})
    classes = {}
    for c, ms in cms.items():
        fl = (kc.get(c) or {}).get("algorithmic_tflop")
        classes[c] = {"ms_per_step_rocprof": round(ms, 4), "ms_per_step_bench_share": (kc.get(c) or {}).get("ms"), "algorithmic_tflop_per_step": fl,
                      "tflops": round(fl / (ms * 1e-3), 1) if fl and ms > 0 else None, "frac_of_2500": round(fl / (ms * 1e-3) / 2500.0, 4) if fl and ms > 0 else None}
    tot_ms = sum(r["ms_per_step"] for r in rows)
    ex = (d.get("roofline_step") or {}).get("executed_tflop_per_step")
    out = {"source": f"rocprofv3 --kernel-trace of `bench.py --steps 20 --warmup 3 --headline-only` ({tag}); {n_iter} loop iterations + 1 eager profiled forward in the trace",
           "bench_line": {"value": d.get("value"), "ms_per_step": d.get("ms_per_step"), "roofline_frac": (d.get("roofline") or {}).get("frac"),
                          "roofline_xattn_frac": (d.get("roofline_xattn") or {}).get("frac"), "roofline_step_frac": (d.get("roofline_step") or {}).get("frac")},
           "traced_kernel_ms_per_step": round(tot_ms, 4), "classes": classes,
           "step": {"executed_tflop_per_step": ex, "tflops_on_traced_kernel_time": round(ex / (tot_ms * 1e-3), 1) if ex and tot_ms else None,
                    "frac_of_2500": round(ex / (tot_ms * 1e-3) / 2500.0, 4) if ex and tot_ms else None},
           "compulsory_bytes_headline_shape": comp, "kernels": rows}
    json.dump(out, open(os.path.join(PROF, f"{tag}_roofline.json"), "w"), indent=1)
    print("roofline:", {c: (v["ms_per_step_rocprof"], v["frac_of_2500"]) for c, v in classes.items()})


if "--aggregate-only" in sys.argv:
    aggregate()
    aggregate_mfma()
elif "--roofline-only" in sys.argv:
    roofline()
else:
    publish()
    roofline()
