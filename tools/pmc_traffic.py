#!/usr/bin/env python3
"""Per-kernel HBM traffic from the rocprofv3 --pmc passes of tools/pmc_run.sh.
usage: pmc_traffic.py <pmc dir> "<workload name as bench.py prints it>" > profiles/<round>_pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are reported in KiB; the value kept is the mean per dispatch * 1024."""
import collections
import csv
import glob
import json
import os
import sys

d, workload = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: collections.defaultdict(float)))
for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] in ("FETCH_SIZE", "WRITE_SIZE"):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[name][row["Counter_Name"]][int(row["Dispatch_Id"])] += float(row["Counter_Value"])
kernels = {}
for name in sorted(acc):
    if not name.startswith("bsp::"):
        continue
    f, w = acc[name]["FETCH_SIZE"], acc[name]["WRITE_SIZE"]
    kernels[name] = {"fetch_bytes": int(sum(f.values()) / max(len(f), 1) * 1024),
                     "write_bytes": int(sum(w.values()) / max(len(w), 1) * 1024),
                     "dispatches": max(len(f), len(w))}
# one multiply = one dispatch of the row-size scan (k_scan_apply<int, false>); a kernel's dispatches per
# step follow from that (the one-off bspgemm_row_work_prefix call of bench.py rounds away)
n_mult = max([v["dispatches"] for k, v in kernels.items() if k.startswith("bsp::k_scan_apply<int, false>")] + [1])
step_total = 0
for k, v in kernels.items():
    v["dispatches_per_step"] = int(round(v["dispatches"] / n_mult))
    step_total += (v["fetch_bytes"] + v["write_bytes"]) * v["dispatches_per_step"]
json.dump({"step_total_bytes": int(step_total), "multiplies_profiled": n_mult, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_run.sh) on `bench.py --steps 2 "
                     "--warmup 1`; KiB*1024, mean per dispatch; FETCH_SIZE not corrected (gfx950 may under-count wide "
                     "coalesced reads by 2x)",
           "workload": workload, "kernels": kernels}, sys.stdout, indent=1)
print()
