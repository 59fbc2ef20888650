"""Times the default flow and its stitch phase (count scan + compaction) on one box, several inputs.
usage: python3 tools/compact_time.py [reps]"""
import sys, time
sys.path.insert(0, "binary-spgemm_amd")
import torch, bspgemm
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
ctx = bspgemm.Context(0)
for name, gen in (("rmat22", lambda: bspgemm.gen_rmat(22, 16, (0.30, 0.25, 0.25), seed=1)),
                  ("uniform18", lambda: bspgemm.gen_uniform(1 << 18, 16, seed=1)),
                  ("g500", lambda: bspgemm.gen_rmat(18, 16, (0.57, 0.19, 0.19), seed=1)),
                  ("powerlaw", lambda: bspgemm.gen_powerlaw(1 << 20, 64, seed=1))):
    rp, ci, n = gen()
    A = ctx.upload(rp, ci, n)
    for _ in range(3):
        ctx.multiply(A, A).free()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(reps):
        C = ctx.multiply(A, A); nnz = C.nnz; C.free()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t) / reps
    st = ctx.stats()
    print("%-10s %.3f ms  %.1f GNZ/s  prepass %.2f numeric %.2f stitch %.3f" %
          (name, dt * 1e3, nnz / dt / 1e9, st["ms_prepass"], st["ms_numeric"], st["ms_stitch"]), flush=True)
    A.free()
