#!/usr/bin/env python3
"""Copies the judged summaries of tools/final_prof_r4.sh's output (gpurun_out/final4, scratch) into profiles/
(tracked), and derives the per-step family timeline from the rocprofv3 kernel trace.
usage: tools/keep_profiles_r2.py [tag]   (tag defaults to r04)"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", os.environ.get("BSPGEMM_PROF_DIR", "final4"))
DST = os.path.join(ROOT, "profiles")
tag = sys.argv[1] if len(sys.argv) > 1 else "r04"


def one(pattern):
    m = sorted(glob.glob(os.path.join(SRC, pattern)))
    return m[0] if m else None


def copy(pattern, name):
    src = one(pattern)
    if src:
        shutil.copy(src, os.path.join(DST, "%s_%s" % (tag, name)))
        return True
    print("missing:", pattern)
    return False


def short(name):
    return name.split("(")[0].replace("void ", "")


def family_timeline(trace_csv, out_txt, what):
    """per multiply (one k_scan_apply<long long, true> dispatch each): first start / last end of every kernel
    family relative to the multiply's first kernel, averaged over the steps -- what shows how the class
    launches of the two streams overlap (the --stats CSV has averages only)"""
    rows = list(csv.DictReader(open(trace_csv)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    steps, cur = [], None
    for r in rows:
        n = short(r["Kernel_Name"])
        if not n.startswith("bsp::"):
            continue
        if n.startswith("bsp::k_row_work") or n.startswith("bsp::k_row_products"):
            cur = []
            steps.append(cur)
        if cur is not None:
            cur.append((n, int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    steps = [s for s in steps if len(s) > 8][1:]          # drop bspgemm_row_work_prefix's lone prepass
    fam = {}
    for s in steps:
        t0 = s[0][1]
        per = {}
        for n, a, b in s:
            key = n.split("<")[0]
            lo, hi, busy, k = per.get(key, (1e30, 0, 0, 0))
            per[key] = (min(lo, a - t0), max(hi, b - t0), busy + (b - a), k + 1)
        for key, v in per.items():
            fam.setdefault(key, []).append(v)
        fam.setdefault("(whole multiply)", []).append((0, max(b for _, _, b in s) - t0, 0, len(s)))
    with open(out_txt, "w") as f:
        f.write("# %s: %d multiplies from %s\n" % (what, len(steps), os.path.basename(trace_csv)))
        f.write("# per kernel family, averaged over the multiplies, ms relative to the multiply's first kernel start:\n")
        f.write("# %-28s %10s %10s %12s %14s %9s\n" % ("family", "first_start", "last_end", "span", "sum_of_durations", "launches"))
        for key in sorted(fam, key=lambda k: sum(v[0] for v in fam[k])):
            v = fam[key]
            m = [sum(x[i] for x in v) / len(v) for i in range(4)]
            f.write("%-30s %10.3f %10.3f %12.3f %14.3f %9.1f\n" % (key, m[0] / 1e6, m[1] / 1e6, (m[1] - m[0]) / 1e6, m[2] / 1e6, m[3]))


os.makedirs(DST, exist_ok=True)
copy("bench_rmat22.json", "bench_rmat22.json")
copy("bench_rmat22_exact.json", "bench_rmat22_exact_flow.json")
if one("dropin.log"):
    copy("dropin.log", "dropin.log")
for flow, d in (("", "stats"), ("exact_flow_", "stats_exact")):
    copy(d + "/*/*kernel_stats.csv", flow + "rmat22_kernel_stats.csv")
    if copy(d + "/*/*kernel_trace.csv", flow + "rmat22_kernel_trace.csv"):
        family_timeline(os.path.join(DST, "%s_%srmat22_kernel_trace.csv" % (tag, flow)),
                        os.path.join(DST, "%s_%srmat22_family_timeline.txt" % (tag, flow)),
                        "bench.py --steps 10 --warmup 3" + (" (BSPGEMM_FLOW=exact)" if flow else ""))
wl = None
for cand in ("bench_rmat22.json", "pmc_ub/fetch.json", "pmc_ub/write.json"):
    try:
        wl = json.loads(open(os.path.join(SRC, cand)).read().strip().splitlines()[-1])["config"]["workload"]
        break
    except Exception:
        continue
for flow, d in (("", "pmc_ub"), ("exact_flow_", "pmc_exact")):
    if wl and os.path.isdir(os.path.join(SRC, d)):
        model = os.path.join(SRC, "gather_model.json")
        args = [sys.executable, os.path.join(ROOT, "tools", "pmc_traffic.py"), os.path.join(SRC, d), wl]
        if os.path.exists(model) and os.path.getsize(model) > 10:
            args.append(model)
        out = subprocess.run(args, capture_output=True, text=True).stdout
        name = "%s_%spmc_traffic.json" % (tag, flow) if not flow else "%s_%spmc.json" % (tag, flow)
        open(os.path.join(DST, name), "w").write(out)
for flow, d in (("", "sq_ub"), ("exact_flow_", "sq_exact")):
    copy(d + "/summary.txt", flow + "pmc_sq_summary.txt")
for f in ("other_upper-bound.jsonl", "other_exact.jsonl", "flows.log", "small.log", "timeline.log", "masked.log",
          "bins_powerlaw.log", "bins_g500.log"):
    copy(f, f.replace("other_", "other_workloads_"))
print("\n".join(sorted(x for x in os.listdir(DST) if x.startswith(tag))))
