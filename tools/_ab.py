import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "binary-spgemm_amd"))
import torch, bspgemm
if len(sys.argv) > 1: bspgemm.LIB_PATH = os.path.abspath(sys.argv[1])
ctx = bspgemm.Context(0)
rp, ci, n = bspgemm.gen_rmat(22, 16, (0.30, 0.25, 0.25), seed=1)
A = ctx.upload(rp, ci, n)
for _ in range(5): ctx.multiply(A, A).free()
torch.cuda.synchronize(); t = time.perf_counter()
acc = {"ms_prepass": 0, "ms_numeric": 0, "ms_stitch": 0}
for _ in range(20):
    C = ctx.multiply(A, A); C.free(); st = ctx.stats()
    for k in acc: acc[k] += st[k] / 20
torch.cuda.synchronize(); dt = (time.perf_counter() - t) / 20
print("%-22s %.3f ms  prepass %.3f numeric %.3f stitch %.3f" % (os.path.basename(sys.argv[1]) if len(sys.argv) > 1 else "tree", dt * 1e3, acc["ms_prepass"], acc["ms_numeric"], acc["ms_stitch"]))
