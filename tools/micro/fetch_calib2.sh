#!/bin/bash
# builds and profiles tools/micro/fetch_calib2.hip on the GPU box:  tools/micro/fetch_calib2.sh <outdir>
OUT=$(realpath -m "$1"); ROOT=$(pwd); mkdir -p "$OUT"
hipcc --offload-arch=gfx950 -O3 -o "$OUT/fetch_calib2" tools/micro/fetch_calib2.hip || exit 1
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_RD[A-Za-z0-9_]*\|TCC_REQ[A-Za-z0-9_]*\|TCC_READ[A-Za-z0-9_]*" | sort -u > "$OUT/tcc_counters.txt"
pass() { # name, counters...
  n=$1; shift
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$OUT/$n" -- "$OUT/fetch_calib2" > "$OUT/$n.log" 2>&1 || echo "pass $n failed"
}
pass fetch FETCH_SIZE
pass rdreq TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass hit TCC_HIT_sum TCC_MISS_sum
timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d "$OUT/time" -- "$OUT/fetch_calib2" > "$OUT/time.log" 2>&1 || echo "pass time failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
names = ["64B-aligned", "32B-aligned", "16B-aligned", "dword-aligned"]
rows = (1 << 30) // 64 // 4
for which in ("fetch", "rdreq", "hit"):
    f = glob.glob(out + "/" + which + "/*/*counter_collection.csv")
    if not f: print("no", which); continue
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f[0])):
        if "k_rows" not in r["Kernel_Name"]: continue
        per[int(r["Dispatch_Id"])][r["Counter_Name"]] = per[int(r["Dispatch_Id"])].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
        per[int(r["Dispatch_Id"])]["_k"] = "L=20" if "<20>" in r["Kernel_Name"] else "L=16"
    ids = sorted(per)
    for i, d in enumerate(ids):
        v = (i // 2) % 4
        L = 16 if per[d]["_k"] == "L=16" else 20
        nrows = rows if L == 16 else rows * 3 // 4
        vals = {k: x for k, x in per[d].items() if k != "_k"}
        txt = "  ".join("%s=%.0f (%.2f per row)" % (k, x, x / nrows) for k, x in sorted(vals.items()))
        if "FETCH_SIZE" in vals:
            txt += "   => %.1f bytes per row (KiB*1024), algorithmic %d + 8" % (vals["FETCH_SIZE"] * 1024 / nrows, L * 4)
        print("%-6s rep %d  %-14s %s  %s" % (which, i // 8, names[v], per[d]["_k"], txt))
PY
python3 - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
names = ["64B-aligned", "32B-aligned", "16B-aligned", "dword-aligned"]
rows = (1 << 30) // 64 // 4
f = glob.glob(out + "/time/*/*kernel_trace.csv")
if f:
    ks = [r for r in csv.DictReader(open(f[0])) if "k_rows" in r["Kernel_Name"]]
    ks.sort(key=lambda r: int(r["Start_Timestamp"]))
    for i, r in enumerate(ks):
        L = 20 if "<20>" in r["Kernel_Name"] else 16
        nrows = rows if L == 16 else rows * 3 // 4
        us = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
        print("time   rep %d  %-14s L=%d  %.1f us  -> %.0f GB/s of algorithmic row bytes" % (i // 8, names[(i // 2) % 4], L, us, nrows * L * 4 / us / 1e3))
PY
rm -f "$OUT/fetch_calib2"; find "$OUT" -name "*.db" -delete
