"""ctypes front-end for the CHECKER libraries.  Test infrastructure only.

* ``liboracle.so``       -- the in-repo CPU restatement (oracle/bspgemm_oracle.c)
* ``_ref/libref_*.so``   -- the untouched reference compiled by oracle/Makefile (present only
                            where /root/reference was available at build time; travels to the
                            GPU box as a prebuilt file)

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product (binary-spgemm_amd/) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_I32P = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_I64P = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")


def build(ref=True):
    """Compile liboracle.so (and oracle/_ref when the reference checkout is present)."""
    subprocess.run(["make", "-C", _HERE, "liboracle.so"] + (["ref"] if ref else []),
                   check=True, stdout=subprocess.DEVNULL)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


_lib = None
_libc = C.CDLL(None)
_libc.malloc.restype = C.c_void_p
_libc.malloc.argtypes = [C.c_size_t]


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(path):
            build(ref=False)
        L = C.CDLL(path)
        L.oracle_bigslice.restype = C.c_int64
        L.oracle_bigslice.argtypes = [_I32P, _I32P, _I32P, _I32P, C.c_int,
                                      C.POINTER(C.POINTER(C.c_int)), _I64P, C.POINTER(C.c_int64),
                                      C.c_int, C.c_int]
        L.oracle_count_products.restype = C.c_int64
        L.oracle_count_products.argtypes = [_I32P, _I32P, _I32P, C.c_int, C.c_int]
        L.oracle_spgemm_omp.restype = C.c_int64
        L.oracle_spgemm_omp.argtypes = [_I32P, C.c_void_p, C.c_int, _I32P, _I32P, C.c_int,
                                        C.POINTER(C.POINTER(C.c_int)), _I64P, C.c_int, C.c_int]
        L.oracle_spgemm.restype = C.c_int64
        L.oracle_spgemm.argtypes = [_I32P, _I32P, C.c_int, _I32P, _I32P, C.c_int,
                                    C.POINTER(C.POINTER(C.c_int)), _I64P]
        L.oracle_spgemm_masked.restype = C.c_int64
        L.oracle_spgemm_masked.argtypes = [_I32P, _I32P, C.c_int, _I32P, _I32P, C.c_int, _I32P, _I32P,
                                           C.POINTER(C.POINTER(C.c_int)), _I64P]
        L.oracle_csr_equal64.restype = C.c_int
        L.oracle_csr_equal64.argtypes = [_I32P, _I64P, _I32P, _I64P, C.c_int]
        L.oracle_readCOO.restype = C.c_int
        L.oracle_readCOO.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_int)), C.POINTER(C.POINTER(C.c_int)),
                                     C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.oracle_free.argtypes = [C.c_void_p]
        _lib = L
    return _lib


def _take(ptr, n):
    """Copy n ints out of a malloc'd buffer and free it."""
    out = np.ctypeslib.as_array(ptr, shape=(max(int(n), 1),))[: int(n)].astype(np.int32, copy=True)
    lib().oracle_free(ptr)
    return out


def spgemm(a_rp, a_ci, b_rp, b_ci, bm):
    """Serial C = A*B (boolean).  Returns (row_ptr int64[n+1], col_idx int32[nnz])."""
    a_rp, a_ci, b_rp, b_ci = map(_i32, (a_rp, a_ci, b_rp, b_ci))
    n = a_rp.size - 1
    crow = np.zeros(n + 1, dtype=np.int64)
    cc = C.POINTER(C.c_int)()
    nnz = lib().oracle_spgemm(a_ci, a_rp, n, b_ci, b_rp, int(bm), C.byref(cc), crow)
    if nnz < 0:
        raise MemoryError("oracle_spgemm")
    return crow, _take(cc, nnz)


def spgemm_rows(a_rp, a_ci, b_rp, b_ci, bm, r0, r1):
    """Rows [r0,r1) only; row_ptr is slice-local (starts at 0) like SpGEMM_bigslice."""
    a_rp, a_ci, b_rp, b_ci = map(_i32, (a_rp, a_ci, b_rp, b_ci))
    crow = np.zeros(r1 - r0 + 1, dtype=np.int64)
    csize = C.c_int64(max(int(bm), 1))
    cc = C.cast(_libc.malloc(csize.value * 4), C.POINTER(C.c_int))
    nnz = lib().oracle_bigslice(a_ci, a_rp, b_ci, b_rp, int(bm), C.byref(cc), crow, C.byref(csize), r0, r1)
    if nnz < 0:
        raise MemoryError("oracle_bigslice")
    return crow, _take(cc, nnz)


def spgemm_omp(a_rp, a_ci, b_rp, b_ci, bm, tblock, threads, row0=0, rows=None):
    """Restated SpGEMM_omp on rows [row0,row0+rows) (interior Arow pointer, like :171)."""
    a_rp, a_ci, b_rp, b_ci = map(_i32, (a_rp, a_ci, b_rp, b_ci))
    n = a_rp.size - 1
    rows = n - row0 if rows is None else rows
    crow = np.zeros(rows + 1, dtype=np.int64)
    cc = C.POINTER(C.c_int)()
    arow_ptr = a_rp.ctypes.data + 4 * row0
    nnz = lib().oracle_spgemm_omp(a_ci, arow_ptr, rows, b_ci, b_rp, int(bm), C.byref(cc), crow,
                                  int(tblock), int(threads))
    if nnz < 0:
        raise MemoryError("oracle_spgemm_omp")
    return crow, _take(cc, nnz)


def spgemm_masked(a_rp, a_ci, b_rp, b_ci, bm, f_rp, f_ci):
    a_rp, a_ci, b_rp, b_ci, f_rp, f_ci = map(_i32, (a_rp, a_ci, b_rp, b_ci, f_rp, f_ci))
    n = a_rp.size - 1
    crow = np.zeros(n + 1, dtype=np.int64)
    cc = C.POINTER(C.c_int)()
    nnz = lib().oracle_spgemm_masked(a_ci, a_rp, n, b_ci, b_rp, int(bm), f_ci, f_rp, C.byref(cc), crow)
    if nnz < 0:
        raise MemoryError("oracle_spgemm_masked")
    return crow, _take(cc, nnz)


def count_products(a_rp, a_ci, b_rp, r0=0, r1=None):
    a_rp, a_ci, b_rp = map(_i32, (a_rp, a_ci, b_rp))
    r1 = a_rp.size - 1 if r1 is None else r1
    return int(lib().oracle_count_products(a_ci, a_rp, b_rp, r0, r1))


def csr_equal(rp1, ci1, rp2, ci2):
    rp1 = np.ascontiguousarray(rp1, dtype=np.int64)
    rp2 = np.ascontiguousarray(rp2, dtype=np.int64)
    if rp1.size != rp2.size:
        return False
    return bool(lib().oracle_csr_equal64(_i32(ci1), rp1, _i32(ci2), rp2, rp1.size - 1))


def read_mtx(path):
    """Restated readCOO: returns (row_ptr int32[M+1], col_idx int32[nnz], M, N) -- transposed CSR."""
    rp, ci = C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
    m, n, nz = C.c_int(), C.c_int(), C.c_int()
    rc = lib().oracle_readCOO(os.fsencode(path), C.byref(rp), C.byref(ci), C.byref(m), C.byref(n), C.byref(nz))
    if rc != 0:
        raise OSError(rc, "oracle_readCOO failed with code %d" % rc)
    return _take(rp, m.value + 1), _take(ci, nz.value), m.value, n.value


# ------------------------------------------------------------------------------------------
# The real reference, compiled by oracle/Makefile into oracle/_ref (never shipped as source).
class Reference:
    """Thin ctypes view of oracle/_ref/libref_final.so and libref_matlab.so."""

    def __init__(self):
        fin = os.path.join(_HERE, "_ref", "libref_final.so")
        mat = os.path.join(_HERE, "_ref", "libref_matlab.so")
        if not (os.path.exists(fin) and os.path.exists(mat)):
            raise FileNotFoundError("oracle/_ref not built (needs /root/reference at build time)")
        self._fin = C.CDLL(fin, mode=os.RTLD_LOCAL)
        self._mat = C.CDLL(mat, mode=os.RTLD_LOCAL)
        IP = C.POINTER(C.c_int)
        # final/SpGEMM_mpi_omp.c:15-18
        self._fin.SpGEMM_bigslice.argtypes = [_I32P, C.c_void_p, C.c_int, _I32P, _I32P, C.c_int,
                                             C.POINTER(IP), _I32P, C.POINTER(C.c_int), C.c_int, C.c_int]
        self._fin.SpGEMM_bigslice.restype = None
        # final/SpGEMM_mpi_omp.c:71-74
        self._fin.SpGEMM_omp.argtypes = [_I32P, C.c_void_p, C.c_int, _I32P, _I32P, C.c_int,
                                        C.POINTER(IP), _I32P, C.c_int]
        self._fin.SpGEMM_omp.restype = None
        # final/SpGEMM_mpi_omp.c:232-235
        self._fin.SpGEMM_masked.argtypes = [_I32P, _I32P, C.c_int, _I32P, _I32P, C.c_int, _I32P, _I32P,
                                           C.POINTER(IP), _I32P, C.POINTER(C.c_int)]
        self._fin.SpGEMM_masked.restype = None
        # final/utils.h:13
        self._fin.readCOO.argtypes = [C.c_char_p, C.POINTER(C.POINTER(C.c_uint32)), C.POINTER(C.POINTER(C.c_uint32)),
                                     C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
        self._fin.readCOO.restype = None
        # Matlab/inc/BSpGEMM.h:2-4
        self._mat.SpGEMM_mat.argtypes = [_I32P, _I32P, C.c_int, _I32P, _I32P, C.c_int, _I32P, _I32P]
        self._mat.SpGEMM_mat.restype = None
        self._libc = C.CDLL(None)
        self._libc.malloc.restype = C.c_void_p
        self._libc.malloc.argtypes = [C.c_size_t]
        self._libc.free.argtypes = [C.c_void_p]

    def _take(self, ptr, n):
        out = np.ctypeslib.as_array(ptr, shape=(max(int(n), 1),))[: int(n)].astype(np.int32, copy=True)
        self._libc.free(C.cast(ptr, C.c_void_p))
        return out

    def bigslice(self, a_rp, a_ci, b_rp, b_ci, bm, r0, r1):
        a_rp, a_ci, b_rp, b_ci = map(_i32, (a_rp, a_ci, b_rp, b_ci))
        crow = np.zeros(r1 - r0 + 1, dtype=np.int32)
        csize = C.c_int(max(int(bm), 1))
        cc = C.cast(self._libc.malloc(csize.value * 4), C.POINTER(C.c_int))
        self._fin.SpGEMM_bigslice(a_ci, a_rp.ctypes.data, a_rp.size - 1, b_ci, b_rp, int(bm),
                                 C.byref(cc), crow, C.byref(csize), r0, r1)
        return crow, self._take(cc, crow[-1])

    def omp(self, a_rp, a_ci, b_rp, b_ci, bm, tblock, row0=0, rows=None):
        """SpGEMM_omp; thread count = OMP_NUM_THREADS / omp default of the process."""
        a_rp, a_ci, b_rp, b_ci = map(_i32, (a_rp, a_ci, b_rp, b_ci))
        n = a_rp.size - 1
        rows = n - row0 if rows is None else rows
        assert rows % tblock == 0, "reference drops remainder rows (final/SpGEMM_mpi_omp.c:77)"
        crow = np.zeros(rows + 1, dtype=np.int32)
        cc = C.POINTER(C.c_int)()
        self._fin.SpGEMM_omp(a_ci, a_rp.ctypes.data + 4 * row0, rows, b_ci, b_rp, int(bm), C.byref(cc), crow, int(tblock))
        return crow, self._take(cc, crow[-1])

    def masked(self, a_rp, a_ci, b_rp, b_ci, bm, f_rp, f_ci):
        a_rp, a_ci, b_rp, b_ci, f_rp, f_ci = map(_i32, (a_rp, a_ci, b_rp, b_ci, f_rp, f_ci))
        n = a_rp.size - 1
        assert bm <= n, "reference sizes xb by An (final/SpGEMM_mpi_omp.c:240)"
        crow = np.zeros(n + 1, dtype=np.int32)
        csize = C.c_int(max(n, 1))
        cc = C.cast(self._libc.malloc(csize.value * 4), C.POINTER(C.c_int))
        self._fin.SpGEMM_masked(a_ci, a_rp, n, b_ci, b_rp, int(bm), f_ci, f_rp, C.byref(cc), crow, C.byref(csize))
        return crow, self._take(cc, crow[-1])

    def spgemm_mat(self, a_rp, a_ci, b_rp, b_ci, bm, nnz_c):
        """SpGEMM_mat with caller-preallocated Ccol of the true nnz (Matlab/SpGEMM.m:4)."""
        a_rp, a_ci, b_rp, b_ci = map(_i32, (a_rp, a_ci, b_rp, b_ci))
        n = a_rp.size - 1
        assert bm <= n, "SpGEMM_mat sizes xb by An (Matlab/inc/BSpGEMM.c:17)"
        crow = np.zeros(n + 1, dtype=np.int32)
        ccol = np.zeros(max(int(nnz_c), 1), dtype=np.int32)
        self._mat.SpGEMM_mat(a_ci, a_rp, n, b_ci, b_rp, int(bm), ccol, crow)
        return crow, ccol[: int(nnz_c)]

    def read_mtx(self, path):
        rp, ci = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
        m, n, nz = C.c_uint32(), C.c_uint32(), C.c_uint32()
        self._fin.readCOO(os.fsencode(path), C.byref(rp), C.byref(ci), C.byref(m), C.byref(n), C.byref(nz))
        return (self._take(C.cast(rp, C.POINTER(C.c_int)), m.value + 1),
                self._take(C.cast(ci, C.POINTER(C.c_int)), nz.value), m.value, n.value)


_ref = None


def reference():
    """The compiled reference, or None where oracle/_ref was not built / cannot load."""
    global _ref
    if _ref is None:
        try:
            _ref = Reference()
        except (OSError, FileNotFoundError):
            _ref = False
    return _ref or None
