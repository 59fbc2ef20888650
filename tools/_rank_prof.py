import os, sys, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "binary-spgemm_amd"))
import numpy as np, torch, bspgemm
ctx = bspgemm.Context(0); ctx.set_class_timing(True); ctx.set_option("class_streams", 1)
rp, ci, n = bspgemm.gen_powerlaw(1 << 20, 64, seed=1)
A = ctx.upload(rp, ci, n)
L = bspgemm.lib()
buf = (C.c_ulonglong * 8)()
reps = 3
for i in range(reps):
    Cc = ctx.multiply(A, A); Cc.free()
st = ctx.stats(); B = st["bins"]
L.bspgemm_debug_rank_prof(buf)
v = np.array(list(buf), dtype=np.float64) / reps
rows = st["rows_per_bin"][B - 3]
print("rank class %.3f ms, %d rows" % (st["ms_bin"][B - 3], rows))
names = ["clear+init", "row record", "sweep 1 (extents, plan, loads, top bits)", "top ranks", "sweep 2 (loads, slot bits)", "read-out to staging", "copy out"]
for k in range(7):
    print("  %-45s %8.0f ticks per row  %5.1f %%" % (names[k], v[k] / rows, 100 * v[k] / v[:7].sum()))
print("  total ticks per row %.0f" % (v[:7].sum() / rows))
