import sys
sys.path.insert(0, "binary-spgemm_amd")
import torch, bspgemm
ctx = bspgemm.Context(0)
ctx.set_class_timing(True)
rp, ci, n = bspgemm.gen_rmat(22, 16, (0.30, 0.25, 0.25), seed=1)
A = ctx.upload(rp, ci, n)
for i in range(4):
    C = ctx.multiply(A, A); C.free()
st = ctx.stats()
print({k: round(st[k], 3) for k in ("ms_total", "ms_prepass", "ms_count", "ms_numeric", "ms_stitch")})
for name, t, d in (("count", st["t_bin_count"], st["ms_bin_count"]), ("numeric", st["t_bin"], st["ms_bin"])):
    for b in range(1, len(t)):
        if d[b] > 0:
            print("%-8s class %2d  start %7.3f  end %7.3f  (%.3f ms)" % (name, b, t[b], t[b] + d[b], d[b]))
