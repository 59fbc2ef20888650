"""Times one equal-work row shard of a big R-MAT product on one GPU (what one rank of an N-GPU run does).
usage: shard_time.py <scale> <parts>"""
import sys, time
sys.path.insert(0, "binary-spgemm_amd")
import torch, bspgemm
scale, parts = int(sys.argv[1]), int(sys.argv[2])
ctx = bspgemm.Context(0)
t = time.perf_counter()
rp, ci, n = bspgemm.gen_rmat(scale, 16, (0.30, 0.25, 0.25), seed=1)
print("generated scale %d in %.1f s" % (scale, time.perf_counter() - t), flush=True)
A = ctx.upload(rp, ci, n)
bounds = ctx.partition_rows(A, A, parts)
r0, r1 = int(bounds[0]), int(bounds[1])
ctx.multiply(A, A, r0, r1).free()
t = time.perf_counter()
for _ in range(3):
    C = ctx.multiply(A, A, r0, r1); nnz = C.nnz; C.free()
dt = (time.perf_counter() - t) / 3
st = ctx.stats()
print("scale %d shard 1/%d rows [%d,%d): %.2f ms  %.1f GNZ/s  symbolic %.2f numeric %.2f stitch %.2f" %
      (scale, parts, r0, r1, dt * 1e3, nnz / dt / 1e9, st["ms_symbolic"], st["ms_numeric"], st["ms_stitch"]), flush=True)
