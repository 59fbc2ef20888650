"""Times products whose rows are all short (the one-wave classes' partial-line stores): uniform n = 2^22 with d = 2, 3, 4
(F ~ 4, 9, 16 per row) and R-MAT scale 20 with edge factor 4.  usage: python3 tools/tiny_rows_time.py"""
import sys
sys.path.insert(0, "binary-spgemm_amd")
import numpy as np
import torch, bspgemm
ctx = bspgemm.Context(0)
cases = [("uniform 2^22 d=%d" % d, (lambda d=d: bspgemm.gen_uniform(1 << 22, d, seed=1))) for d in (2, 3, 4)]
cases.append(("rmat 20 ef=4", lambda: bspgemm.gen_rmat(20, 4, (0.30, 0.25, 0.25), seed=1)))
for name, gen in cases:
    rp, ci, n = gen()
    A = ctx.upload(rp, ci, n)
    ts = []
    for r in range(12):
        C = ctx.multiply(A, A); st = ctx.stats(); nnz = C.nnz; C.free()
        if r >= 2: ts.append((st["ms_total"], st["ms_prepass"], st["ms_numeric"], st["ms_stitch"]))
    t = np.median(np.array(ts), axis=0)
    print("%-18s total %.3f prepass %.3f numeric %.3f stitch %.3f ms  nnz %d" % (name, t[0], t[1], t[2], t[3], nnz), flush=True)
    A.free()
