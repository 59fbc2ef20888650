#!/usr/bin/env python3
"""Per-kernel register / LDS / occupancy table of one HIP source (hipcc -Rpass-analysis).
    tools/kres.py binary-spgemm_amd/csrc/wave_count.hip [-D...]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-Iinclude", "-I../include",
       "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in err.splitlines():
    m = re.search(r"remark: +([\w \[\]/]+): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1).strip(), m.group(2)
    if k in ("Function Name", "Name"):
        cur = {"name": subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().split("(")[0]}
        rows.append(cur)
    else:
        cur[k] = v
print("%-60s %5s %5s %5s %6s %8s %7s" % ("kernel", "SGPR", "VGPR", "AGPR", "occ", "LDS", "scratch"))
for r in rows:
    print("%-60s %5s %5s %5s %6s %8s %7s" % (r["name"][-60:], r.get("TotalSGPRs", r.get("SGPRs", "?")), r.get("VGPRs", "?"), r.get("AGPRs", "?"),
                                          r.get("Occupancy [waves/SIMD]", "?"), r.get("LDS Size [bytes/block]", "?"),
                                          r.get("ScratchSize [bytes/lane]", "?")))
