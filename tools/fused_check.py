"""Development check of the fused flow against the oracle on a ladder of sizes (GPU box).
usage: python tools/fused_check.py [max_scale]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "binary-spgemm_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch  # noqa
import bspgemm
from oracle import oracle as O

max_scale = int(sys.argv[1]) if len(sys.argv) > 1 else 16
ctx = bspgemm.Context(0)

def check(name, rp, ci, n, cols=None):
    cols = cols or n
    A = ctx.upload(rp, ci, cols)
    erp, eci = O.spgemm(rp, ci, rp, ci, cols) if n <= (1 << 17) else (None, None)
    out = {}
    for flow in ("upper-bound", "fused"):
        ctx.set_flow(flow)
        t0 = time.time()
        C = ctx.multiply(A, A)
        dt = time.time() - t0
        crp, cci = C.download()
        st = ctx.stats()
        C.free()
        out[flow] = (crp, cci, st, dt)
    urp, uci, ust, _ = out["upper-bound"]
    frp, fci, fst, fdt = out["fused"]
    ok = np.array_equal(urp, frp) and np.array_equal(uci, fci)
    if erp is not None:
        ok = ok and np.array_equal(erp, frp) and np.array_equal(eci, fci)
    print("%-28s n=%-8d nnzC=%-11d fused %s  (ub %.3f ms, fused %.3f ms, numeric %.3f)" % (
        name, n, frp[-1], "OK" if ok else "MISMATCH", ust["ms_total"], fst["ms_total"], fst["ms_numeric"]), flush=True)
    if not ok:
        bad = np.flatnonzero(urp != frp)
        print("  row_ptr first diffs at", bad[:8], urp[bad[:4]], frp[bad[:4]])
        if urp[-1] == frp[-1]:
            badc = np.flatnonzero(uci != fci)
            print("  col_idx diffs:", badc.size, badc[:8], uci[badc[:4]], fci[badc[:4]])
        sys.exit(1)

rp, ci, m, n = bspgemm.readCOO(os.path.join(ROOT, "tests", "golden", "validity_test.mtx"))
check("validity", rp, ci, n)
for n_, d in ((64, 3), (1000, 5), (4096, 8), (8192, 16), (1 << 14, 16)):
    rp, ci, _ = bspgemm.gen_uniform(n_, d, seed=3)
    check("uniform d=%d" % d, rp, ci, n_)
for sc in range(8, max_scale + 1, 2):
    rp, ci, n_ = bspgemm.gen_rmat(sc, 16, (0.30, 0.25, 0.25), seed=1)
    check("rmat mild s%d" % sc, rp, ci, n_)
for sc in (10, 12, 14):
    if sc <= max_scale:
        rp, ci, n_ = bspgemm.gen_rmat(sc, 16, (0.57, 0.19, 0.19), seed=1)
        check("rmat g500 s%d" % sc, rp, ci, n_)
ctx.close()
print("all ok")
