#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc counter_collection CSVs per kernel: mean counter value per dispatch.
usage: pmc_summary.py <dir> [<dir> ...]   (prints a table; FETCH_SIZE/WRITE_SIZE are in KiB)"""
import collections
import csv
import glob
import os
import sys

acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                name = row["Kernel_Name"].split("(")[0].replace("void ", "")
                acc[name][row["Counter_Name"]].append((int(row["Dispatch_Id"]), float(row["Counter_Value"])))
for name in sorted(acc):
    print(name)
    for cname in sorted(acc[name]):
        per_dispatch = collections.defaultdict(float)
        for did, v in acc[name][cname]:
            per_dispatch[did] += v
        vals = list(per_dispatch.values())
        print("    %-28s mean/dispatch %.6g   (n=%d)" % (cname, sum(vals) / len(vals), len(vals)))
