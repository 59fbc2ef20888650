#!/usr/bin/env python3
"""Does the dword alignment of B's rows cost the numeric kernel anything?  
B = A = a synthetic matrix whose rows ALL have exactly 16 entries (n = 2^22, stratified columns: sorted, distinct,
spread over all of [0, n)), so that every B row is one 64-byte piece and every A row has exactly 256 products (one
capacity class).  The same CSR is multiplied with B.col_idx placed at byte offsets 0 (every row 64-byte aligned), 4,
32 and 60 inside its allocation; the phase times of 10 multiplies are printed per placement.
usage: python tools/align_probe.py [log2 n]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "binary-spgemm_amd"))
import torch  # noqa: E402
import bspgemm  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
n, d = 1 << scale, 16
rng = np.random.default_rng(5)
stride = n // d
ci = (np.arange(d, dtype=np.int64)[None, :] * stride + rng.integers(0, stride, size=(n, d))).astype(np.int32).reshape(-1)
rp = (np.arange(n + 1, dtype=np.int64) * d).astype(np.int32)
dev = torch.device("cuda", 0)
ctx = bspgemm.Context(0)
d_rp = torch.from_numpy(rp).to(dev)
pad = 64
d_buf = torch.zeros(ci.size + 2 * pad, dtype=torch.int32, device=dev)
base = d_buf.data_ptr()
assert base % 256 == 0, base
A = ctx.upload(rp, ci, n)
ref = None
for off_bytes in (0, 4, 32, 60, 0):
    k = off_bytes // 4
    d_buf[k:k + ci.size] = torch.from_numpy(ci).to(dev)
    torch.cuda.synchronize()
    B = ctx.wrap_device(n, n, int(ci.size), d_rp.data_ptr(), base + off_bytes, keep=(d_rp, d_buf))
    for _ in range(3):
        ctx.multiply(A, B).free()
    ms = {"ms_total": [], "ms_prepass": [], "ms_numeric": [], "ms_stitch": []}
    for _ in range(10):
        C = ctx.multiply(A, B)
        st = ctx.stats()
        for key in ms:
            ms[key].append(st[key])
        nnz = C.nnz
        if ref is None:
            ref = C.download()[0]
        C.free()
    B.free()
    print("B.col_idx at +%2d bytes: total %.3f  prepass %.3f  numeric %.3f  stitch %.3f ms   (median of 10; nnz(C) = %d, products = %d)"
          % (off_bytes, *(float(np.median(ms[key])) for key in ("ms_total", "ms_prepass", "ms_numeric", "ms_stitch")), nnz, st["products"]), flush=True)
ctx.close()
