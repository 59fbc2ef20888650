"""Development timing of one flow on one generated input (GPU box).
usage: python tools/fused_time.py flow scale [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "binary-spgemm_amd")):
    sys.path.insert(0, p)
import numpy as np
import torch  # noqa
import bspgemm
flow, scale = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = bspgemm.Context(0)
if scale == "validity":      # BASELINE config 1
    rp, ci, n, _ = bspgemm.readCOO(os.path.join(ROOT, "tests", "golden", "validity_test.mtx"))
elif scale.startswith("u"):  # uNN: uniform n = 2^NN, d = 16 (u18 = BASELINE config 2)
    rp, ci, n = bspgemm.gen_uniform(1 << int(scale[1:]), 16, seed=1)
elif scale.startswith("g"):  # gNN: Graph500-skew R-MAT
    rp, ci, n = bspgemm.gen_rmat(int(scale[1:]), 16, (0.57, 0.19, 0.19), seed=1)
elif scale == "powerlaw":    # BASELINE config 5
    rp, ci, n = bspgemm.gen_powerlaw(1 << 20, 64, 2.1, (1 << 20) // 16, seed=1)
else:
    rp, ci, n = bspgemm.gen_rmat(int(scale), 16, (0.30, 0.25, 0.25), seed=1)
A = ctx.upload(rp, ci, n)
ctx.set_flow(flow)
ts = []
for r in range(reps + 2):
    C = ctx.multiply(A, A)
    st = ctx.stats()
    nnz = C.nnz
    C.free()
    if r >= 2:
        ts.append((st["ms_total"], st["ms_prepass"], st["ms_count"], st["ms_numeric"], st["ms_stitch"]))
t = np.median(np.array(ts), axis=0)
print("%s scale %s env[%s]: total %.3f prepass %.3f count %.3f numeric %.3f stitch %.3f ms  nnz %d  -> %.1f GNZ/s" % (
    flow, scale, " ".join("%s=%s" % (k, v) for k, v in os.environ.items() if k.startswith("BSPGEMM_")), t[0], t[1], t[2], t[3], t[4], nnz, nnz / t[0] / 1e6))
ctx.close()
