"""PCIe-inclusive rate of the int32 drop-in (host arrays in, malloc'ed host arrays out), like the
reference's SpGEMM_omp call site: usage dropin_time.py <scale>"""
import sys, time
sys.path.insert(0, "binary-spgemm_amd")
import torch, bspgemm
scale = int(sys.argv[1])
rp, ci, n = bspgemm.gen_rmat(scale, 16, (0.30, 0.25, 0.25), seed=1)
import ctypes as C
import numpy as np
L = bspgemm.lib()
libc = C.CDLL(None)
libc.free.argtypes = [C.c_void_p]
ci = np.ascontiguousarray(ci, dtype=np.int32)
rp = np.ascontiguousarray(rp, dtype=np.int32)
crow = np.zeros(n + 1, dtype=np.int32)


def call():
    cc = C.POINTER(C.c_int)()
    t = time.perf_counter()
    st = L.SpGEMM_hip(ci, C.c_void_p(rp.ctypes.data), n, ci, rp, n, C.byref(cc), crow, 0)
    dt = time.perf_counter() - t
    assert st == 0, st
    libc.free(C.cast(cc, C.c_void_p))
    return dt


call()                                               # warm-up (context creation, first touch)
best = min(call() for _ in range(3))
nnz = int(crow[-1])
print("scale %d drop-in SpGEMM_hip (upload A,B + multiply + download C into malloc'ed memory): %.1f ms, nnz(C)=%d, %.2f GNZ/s" %
      (scale, best * 1e3, nnz, nnz / best / 1e9))
