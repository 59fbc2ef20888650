#!/usr/bin/env python3
"""Per-kernel HBM traffic from the rocprofv3 --pmc passes of tools/pmc_run.sh.
usage: pmc_traffic.py <pmc dir> "<workload name as bench.py prints it>" > profiles/<round>_pmc_traffic.json
FETCH_SIZE / WRITE_SIZE are reported in KiB; the value kept is the mean per dispatch * 1024.
Correction (MI355X_MICROARCH.md, HBM: FETCH_SIZE tallies 128-byte requests at 64 bytes; calibrated for
this library's access widths with tools/micro/fetch_calib.hip -> profiles/r02_fetch_calibration.txt):
  coalesced streams of 4, 8 or 16 bytes per lane read 0.500 of their bytes  -> FETCH_SIZE x 2
  random 64-byte rows (dword gathers) read 1.03 of their bytes              -> FETCH_SIZE x 1
  WRITE_SIZE is exact for 4- and 16-byte-per-lane stores.
A streaming kernel's fetch is doubled; a gather kernel mixes both kinds of request (B rows straddle
64-byte sectors at random), so its fetch is bracketed: x1 (low) .. x2 (high).
Round 4: with a third argument -- the JSON of tools/gather_traffic_model.py for the same workload -- the kernels that gather
B.col_idx also get `fetch_bytes_calibrated`: the calibration of profiles/r04_fetch_calibration_unaligned.txt (a request is a
128-byte line's wanted sectors, tallied as 64 bytes) applied to the matrix's own row_ptr says how many bytes the gather moves
(gm) and how many FETCH_SIZE shows for it (gt); what the family fetched beyond gt is its coalesced streams (extents, records),
tallied at half: calibrated = FETCH_SIZE * (gm + 2 st) / (gt + st), st = max(sum FETCH_SIZE - gt, 0)."""
import collections
import csv
import glob
import json
import os
import sys

d, workload = sys.argv[1], sys.argv[2]
gmodel = json.load(open(sys.argv[3])) if len(sys.argv) > 3 else None
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
    fb = int(sum(f.values()) / max(len(f), 1) * 1024)
    gather = any(t in name for t in ("k_wave_rows", "k_wave_count", "k_wave_masked", "k_dense_rows", "k_row_work",
                                     "k_row_products", "k_extents_of_rows"))
    kernels[name] = {"fetch_bytes": fb, "fetch_bytes_corrected_low": fb if gather else 2 * fb,
                     "fetch_bytes_corrected_high": 2 * fb, "access": "gather" if gather else "stream",
                     "write_bytes": int(sum(w.values()) / max(len(w), 1) * 1024),
                     "dispatches": max(len(f), len(w))}
if gmodel:
    bcol = [k for k in kernels if any(t in k for t in ("k_wave_rows", "k_dense_rows", "k_wave_masked"))]
    n_mult_g = max([kernels[k]["dispatches"] for k in kernels if k.startswith("bsp::k_scan_apply<int, false>")] + [1])
    shown = sum(kernels[k]["fetch_bytes"] * max(int(round(kernels[k]["dispatches"] / n_mult_g)), 1) for k in bcol)
    gm, gt = gmodel["gather_bytes_moved"], gmodel["gather_bytes_fetch_size_shows"]
    st = max(shown - gt, 0)
    factor = (gm + 2 * st) / (gt + st) if gt + st > 0 else 1.0
    for k in bcol:
        kernels[k]["fetch_bytes_calibrated"] = int(kernels[k]["fetch_bytes"] * factor)
# one multiply = one dispatch of the row-size scan (k_scan_apply<int, false>); a kernel's dispatches per
# step follow from that (the one-off bspgemm_row_work_prefix call of bench.py rounds away)
n_mult = max([v["dispatches"] for k, v in kernels.items() if k.startswith("bsp::k_scan_apply<int, false>")] + [1])
step_total = step_lo = step_hi = step_cal = 0
for k, v in kernels.items():
    v["dispatches_per_step"] = int(round(v["dispatches"] / n_mult))
    step_total += (v["fetch_bytes"] + v["write_bytes"]) * v["dispatches_per_step"]
    step_lo += (v["fetch_bytes_corrected_low"] + v["write_bytes"]) * v["dispatches_per_step"]
    step_hi += (v["fetch_bytes_corrected_high"] + v["write_bytes"]) * v["dispatches_per_step"]
    step_cal += (v.get("fetch_bytes_calibrated", v["fetch_bytes_corrected_high"]) + v["write_bytes"]) * v["dispatches_per_step"]
json.dump({"step_total_bytes": int(step_hi), "step_total_bytes_low": int(step_lo), "step_total_bytes_uncorrected": int(step_total),
           "step_total_bytes_calibrated": int(step_cal) if gmodel else None,
           "gather_model": gmodel, "gather_calibration_factor": round(factor, 4) if gmodel else None,
           "multiplies_profiled": n_mult, "source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, tools/pmc_run.sh) on `bench.py --steps 2 "
                     "--warmup 1`; KiB*1024, mean per dispatch; FETCH_SIZE corrected for gfx950 (128-byte requests tallied at 64): "
                     "x2 for streaming kernels, bracketed x1..x2 for gather kernels (step_total_bytes is the HIGH bound); "
                     "calibration: profiles/r02_fetch_calibration.txt",
           "workload": workload, "kernels": kernels}, sys.stdout, indent=1)
print()
