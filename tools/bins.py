"""Per-class launch timeline of one multiply for a named workload (HIP events of the stats ring).
usage: python3 tools/bins.py {rmat22|g500|powerlaw|uniform} [flow]"""
import sys
sys.path.insert(0, "binary-spgemm_amd")
import numpy as np
import torch, bspgemm
which = sys.argv[1] if len(sys.argv) > 1 else "powerlaw"
ctx = bspgemm.Context(0)
ctx.set_class_timing(True)
if len(sys.argv) > 2:
    ctx.set_flow(sys.argv[2])
if which == "rmat22":
    rp, ci, n = bspgemm.gen_rmat(22, 16, (0.30, 0.25, 0.25), seed=1)
elif which == "g500":
    rp, ci, n = bspgemm.gen_rmat(18, 16, (0.57, 0.19, 0.19), seed=1)
elif which == "uniform":
    rp, ci, n = bspgemm.gen_uniform(1 << 18, 16, seed=1)
else:
    rp, ci, n = bspgemm.gen_powerlaw(1 << 20, 64, seed=1)
A = ctx.upload(rp, ci, n)
for i in range(4):
    C = ctx.multiply(A, A); nnz = C.nnz; C.free()
st = ctx.stats()
print(which, "nnzC", nnz, {k: round(st[k], 3) for k in ("ms_total", "ms_prepass", "ms_count", "ms_numeric", "ms_stitch")})
rows = st.get("rows_per_bin"); prods = None
for name, t, d in (("count", st["t_bin_count"], st["ms_bin_count"]), ("numeric", st["t_bin"], st["ms_bin"])):
    for b in range(1, len(t)):
        if d[b] > 0:
            extra = ""
            if rows is not None: extra = "  rows %9d" % rows[b]
            if prods is not None: extra += "  products %12d" % prods[b]
            print("%-8s class %2d  start %7.3f  end %7.3f  (%.3f ms)%s" % (name, b, t[b], t[b] + d[b], d[b], extra))
