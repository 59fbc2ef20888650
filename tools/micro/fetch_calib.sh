#!/bin/bash
# builds and profiles tools/micro/fetch_calib.hip on the GPU box:  tools/micro/fetch_calib.sh <outdir>
OUT=$(realpath -m "$1"); ROOT=$(pwd); mkdir -p "$OUT"
hipcc --offload-arch=gfx950 -O3 -o "$OUT/fetch_calib" tools/micro/fetch_calib.hip || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/fetch" -- "$OUT/fetch_calib" > "$OUT/fetch.log" 2>&1
timeout -k 10 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/write" -- "$OUT/fetch_calib" > "$OUT/write.log" 2>&1
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for which in ("fetch", "write"):
    f = glob.glob(out + "/" + which + "/*/*counter_collection.csv")
    if not f: print("no", which); continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        acc[(r["Kernel_Name"].split("(")[0], r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(acc.items()):
        print("%-40s %-11s mean %.1f KiB = %.3f of 1 GiB  (n=%d)" % (k, c, sum(v) / len(v), sum(v) / len(v) * 1024 / 2**30, len(v)))
PY
rm -f "$OUT/fetch_calib"; find "$OUT" -name "*.db" -delete
