#!/usr/bin/env python3
"""A/B of the one-wave numeric kernels on one generated input: rank bitmap (default) against the bucket accumulator
(BSPGEMM_OPT_BUCKET_PATH).  Per variant: the phase times of the default two-stream schedule (median of `steps`
multiplies) and, with the class launches serialised on ONE stream and bracketed by events, every class's own time.
usage: python tools/ab_numeric.py [rmat|uniform|g500|powerlaw] [scale] [steps]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "binary-spgemm_amd"))
import bspgemm  # noqa: E402

kind = sys.argv[1] if len(sys.argv) > 1 else "rmat"
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 22
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
if kind == "rmat":
    rp, ci, n = bspgemm.gen_rmat(scale, 16, (0.30, 0.25, 0.25), seed=1)
elif kind == "g500":
    rp, ci, n = bspgemm.gen_rmat(scale, 16, (0.57, 0.19, 0.19), seed=1)
elif kind == "powerlaw":
    rp, ci, n = bspgemm.gen_powerlaw(1 << scale, 64, seed=1)
else:
    rp, ci, n = bspgemm.gen_uniform(1 << scale, 16, seed=1)
ctx = bspgemm.Context(0)
A = ctx.upload(rp, ci, n)
ref = None
for label, bucket in (("rank-bitmap", 0), ("bucket", 1), ("rank-bitmap", 0), ("bucket", 1)):
    ctx.set_option("bucket_path", bucket)
    ctx.set_option("class_streams", 2)
    ctx.set_class_timing(False)
    for _ in range(3):
        ctx.multiply(A, A).free()
    ph = {k: [] for k in ("ms_total", "ms_prepass", "ms_numeric", "ms_stitch")}
    for _ in range(steps):
        C = ctx.multiply(A, A)
        st = ctx.stats()
        for k in ph:
            ph[k].append(st[k])
        if ref is None:
            ref = (C.nnz, C.download(col_idx=False)[0])
        else:
            assert C.nnz == ref[0] and np.array_equal(C.download(col_idx=False)[0], ref[1]), "row_ptr differs between variants"
        C.free()
    print("%-12s two streams: total %.3f  prepass %.3f  numeric %.3f  stitch %.3f ms  (%.1f GNZ/s)"
          % (label, *(float(np.median(ph[k])) for k in ("ms_total", "ms_prepass", "ms_numeric", "ms_stitch")),
             ref[0] / float(np.median(ph["ms_total"])) / 1e6), flush=True)
    ctx.set_option("class_streams", 1)
    ctx.set_class_timing(True)
    per = []
    for _ in range(max(steps // 2, 3)):
        C = ctx.multiply(A, A)
        st = ctx.stats()
        per.append(st["ms_bin"])
        C.free()
    med = np.median(np.array(per), axis=0)
    print("%-12s one stream, per class (cap: rows: ms): %s   sum %.3f"
          % (label, "  ".join("%d:%d:%.3f" % (c, r, m) for c, r, m in zip(st["bin_cap"], st["rows_per_bin"], med) if r and c), float(med.sum())), flush=True)
ctx.close()
