"""Times both multiply flows on one box: tools/flows.py [scale] (R-MAT mild, A*A)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "binary-spgemm_amd"))
import torch, bspgemm
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
ctx = bspgemm.Context(0)
rp, ci, n = bspgemm.gen_rmat(scale, 16, (0.30, 0.25, 0.25), seed=1)
A = ctx.upload(rp, ci, n)
for flow in ("upper-bound", "exact", "upper-bound", "exact"):
    ctx.set_flow(flow)
    for _ in range(3):
        ctx.multiply(A, A).free()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        C = ctx.multiply(A, A); nnz = C.nnz; C.free()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / 10
    st = ctx.stats()
    print("%-12s %.3f ms  %.1f GNZ/s  prepass %.2f count %.2f numeric %.2f stitch %.2f" %
          (flow, dt * 1e3, nnz / dt / 1e9, st["ms_prepass"], st["ms_count"], st["ms_numeric"], st["ms_stitch"]), flush=True)
