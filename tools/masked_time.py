import sys, time
sys.path.insert(0, "binary-spgemm_amd")
import torch, bspgemm
ctx = bspgemm.Context(0)
ctx.set_class_timing(True)
for scale, abc in ((20, (0.30, 0.25, 0.25)), (22, (0.30, 0.25, 0.25)), (18, (0.57, 0.19, 0.19))):
    rp, ci, n = bspgemm.gen_rmat(scale, 16, abc, seed=1)
    A = ctx.upload(rp, ci, n)
    for name, fn in (("A*A", lambda: ctx.multiply(A, A)), ("A.*(A*A)", lambda: ctx.multiply_masked(A, A, A))):
        fn().free()
        t = time.perf_counter()
        for _ in range(5):
            C = fn(); nnz = C.nnz; C.free()
        dt = (time.perf_counter() - t) / 5
        st = ctx.stats()
        print("scale %d %s %-9s %.2f ms  products %.3g  nnz %.3g  bins %s" % (scale, abc, name, dt * 1e3, st["products"], nnz, st["rows_per_bin"]))
        print("    ms: symbolic %.3f numeric %.3f stitch %.3f per-bin %s" % (st["ms_symbolic"], st["ms_numeric"], st["ms_stitch"], [round(x, 3) for x in st["ms_bin"]]))
    A.free()
