"""Wall time and phase split of the small BASELINE shapes (cfg1 validity fixture, cfg2 uniform n=2^18 d=16)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "binary-spgemm_amd"))
import torch, bspgemm
ctx = bspgemm.Context(0)
cases = []
rp, ci, m, n = bspgemm.readCOO(os.path.join(ROOT, "tests", "golden", "validity_test.mtx"))
cases.append(("cfg1 validity", rp, ci, n))
rp, ci, n = bspgemm.gen_uniform(1 << 18, 16, seed=1)
cases.append(("cfg2 uniform 2^18 d16", rp, ci, n))
for name, rp, ci, n in cases:
    A = ctx.upload(rp, ci, n)
    for flow in ("upper-bound", "exact"):
        ctx.set_flow(flow)
        for _ in range(5):
            ctx.multiply(A, A).free()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(50):
            C = ctx.multiply(A, A); nnz = C.nnz; C.free()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 50
        st = ctx.stats()
        print("%-22s %-12s %.3f ms wall  (events: total %.3f prepass %.3f count %.3f numeric %.3f stitch %.3f)  nnz %d  %.1f GNZ/s" %
              (name, flow, dt * 1e3, st["ms_total"], st["ms_prepass"], st["ms_count"], st["ms_numeric"], st["ms_stitch"], nnz, nnz / dt / 1e9), flush=True)
