"""Heavy-row classes timed alone (one class stream), optionally through another build of the library.
usage: python3 tools/heavy_abl.py <workload: g500|g500_20|g500_22|powerlaw> [path of libbspgemm.so]
Timing tool: with an ablated build the RESULT is wrong on purpose; only the class durations mean anything."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "binary-spgemm_amd"))
import numpy as np
import torch, bspgemm
which = sys.argv[1]
if len(sys.argv) > 2:
    bspgemm.LIB_PATH = os.path.abspath(sys.argv[2])
ctx = bspgemm.Context(0)
ctx.set_class_timing(True)
ctx.set_option("class_streams", 1)
if which == "g500":
    rp, ci, n = bspgemm.gen_rmat(18, 16, (0.57, 0.19, 0.19), seed=1)
elif which == "g500_20":
    rp, ci, n = bspgemm.gen_rmat(20, 8, (0.57, 0.19, 0.19), seed=1)
elif which == "g500_22":
    rp, ci, n = bspgemm.gen_rmat(22, 4, (0.57, 0.19, 0.19), seed=1)
else:
    rp, ci, n = bspgemm.gen_powerlaw(1 << 20, 64, seed=1)
A = ctx.upload(rp, ci, n)
for i in range(3):
    C = ctx.multiply(A, A); nnz = C.nnz; C.free()
st = ctx.stats()
d = st["ms_bin"]; rows = st["rows_per_bin"]
B = st["bins"]
print("%-9s %-16s nnzC %11d numeric %7.3f  rank %7.3f ms / %6d rows  mid %7.3f ms / %6d rows   hub %7.3f ms / %5d rows" % (
    which, os.path.basename(sys.argv[2]) if len(sys.argv) > 2 else "product", nnz, st["ms_numeric"], d[B - 3], rows[B - 3], d[B - 2], rows[B - 2],
    d[B - 1], rows[B - 1]), flush=True)
