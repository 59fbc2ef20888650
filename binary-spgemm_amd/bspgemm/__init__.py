"""Python view of libbspgemm.so (ctypes over the C ABI in include/bspgemm.h).

This module is plumbing for tests and bench.py: every compute call goes through the C ABI into
the HIP kernels.  There is no Python or CPU implementation of the product here -- if the shared
library is missing or no gfx950 device is visible the calls raise.

Mirrors the reference's interface names where they exist:
    SpGEMM_hip(...)        <-> SpGEMM_omp      final/SpGEMM_mpi_omp.c:71-74
    SpGEMM_hip_bigslice    <-> SpGEMM_bigslice final/SpGEMM_mpi_omp.c:15-18
    SpGEMM_hip_mat         <-> SpGEMM_mat      Matlab/inc/BSpGEMM.h:2-4
    readCOO                <-> readCOO         final/utils.c:47-81
    csr_equal              <-> SpGEMM_valid    final/SpGEMM_mpi_omp_validity.c:290-302
"""
import ctypes as C
import os
import subprocess
import weakref

import numpy as np

# host-side OpenMP (generators): stay inside one GPU's CPU share on shared boxes
os.environ.setdefault("OMP_NUM_THREADS", str(min(16, len(os.sched_getaffinity(0)))))
# The library runs its class launches on two HIP streams next to its main one; the ROCm runtime
# multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues (default 4).  Once RCCL
# and torch have created theirs, two of the library's streams end up on ONE queue and the class
# launches serialise (+9 % accumulate time, measured).  Read by the HIP runtime when it starts.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

_PKG = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG)                      # binary-spgemm_amd/
LIB_PATH = os.path.join(_ROOT, "libbspgemm.so")

_I32P = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_I64P = np.ctypeslib.ndpointer(dtype=np.int64, flags="C_CONTIGUOUS")

STATUS = {0: "OK", 1: "ERR_INVALID", 2: "ERR_ALLOC", 3: "ERR_HIP", 4: "ERR_NO_DEVICE",
          5: "ERR_OVERFLOW", 6: "ERR_IO", 7: "ERR_FORMAT", 8: "ERR_COMM", 9: "ERR_SIZE"}

# every symbol include/bspgemm.h declares (checked by tests/test_abi.py)
EXPORTS = [
    "bspgemm_status_string", "bspgemm_last_error", "bspgemm_create", "bspgemm_destroy",
    "bspgemm_set_stream", "bspgemm_synchronize", "bspgemm_matrix_upload", "bspgemm_matrix_wrap_device",
    "bspgemm_matrix_free", "bspgemm_matrix_rows", "bspgemm_matrix_cols", "bspgemm_matrix_nnz",
    "bspgemm_multiply", "bspgemm_multiply_masked", "bspgemm_result_rows", "bspgemm_result_nnz",
    "bspgemm_result_row_ptr_device", "bspgemm_result_col_idx_device", "bspgemm_result_download",
    "bspgemm_result_free", "bspgemm_row_work_prefix", "bspgemm_partition_rows", "bspgemm_last_stats",
    "SpGEMM_hip", "SpGEMM_hip_bigslice", "SpGEMM_hip_mat", "SpGEMM_hip_masked", "bspgemm_dropin_set_device",
    "bspgemm_comm_unique_id", "bspgemm_comm_create", "bspgemm_comm_destroy", "bspgemm_comm_stitch_row_ptr", "bspgemm_lengths_to_row_ptr",
    "bspgemm_readCOO", "bspgemm_write_mtx", "bspgemm_write_result_mtx", "bspgemm_csr_equal",
    "bspgemm_csr_equal64", "bspgemm_gen_uniform", "bspgemm_gen_rmat", "bspgemm_gen_powerlaw",
    "bspgemm_matrix_from_result", "bspgemm_closure",
    "bspgemm_readCOO_ex", "bspgemm_comm_create_host", "bspgemm_comm_rank", "bspgemm_comm_size",
    "bspgemm_comm_gather_col_idx", "SpGEMM_hip_multi", "bspgemm_device_count", "bspgemm_stats_at",
    "bspgemm_set_flow", "bspgemm_set_class_timing", "bspgemm_build_info", "bspgemm_matrix_invalidate", "bspgemm_comm_agree", "bspgemm_comm_inject_failure",
    "bspgemm_set_option", "bspgemm_get_option", "bspgemm_matrix_uses_blocked_table", "bspgemm_matrix_uses_padded_rows",
]


class BspgemmError(RuntimeError):
    def __init__(self, status, where=""):
        self.status = status
        msg = lib().bspgemm_last_error().decode(errors="replace") if _lib is not None else ""
        super().__init__("%s: %s %s" % (where, STATUS.get(status, status), msg))


MAX_BINS = 20      # BSPGEMM_MAX_BINS
FLOWS = {"auto": 0, "upper-bound": 1, "exact": 2}                                    # BSPGEMM_FLOW_*
OPTIONS = {"class_streams": 1, "blocked_extents": 2, "check": 3, "small_path": 4, "padded_rows": 5}    # bspgemm_option


class Stats(C.Structure):
    _fields_ = [("rows", C.c_int64), ("nnz_a", C.c_int64), ("products", C.c_int64), ("nnz_c", C.c_int64),
                ("bytes_alg", C.c_int64), ("bytes_read_alg", C.c_int64), ("rows_per_bin", C.c_int64 * MAX_BINS),
                ("ms_total", C.c_float), ("ms_symbolic", C.c_float), ("ms_prepass", C.c_float),
                ("ms_count", C.c_float), ("ms_numeric", C.c_float), ("ms_stitch", C.c_float),
                ("ms_bin", C.c_float * MAX_BINS), ("ms_bin_count", C.c_float * MAX_BINS),
                ("t_bin", C.c_float * MAX_BINS), ("t_bin_count", C.c_float * MAX_BINS), ("bins", C.c_int),
                ("bin_cap", C.c_int * MAX_BINS), ("flow", C.c_int), ("prepass_kernel", C.c_int),
                ("class_streams", C.c_int), ("small_path", C.c_int), ("checked", C.c_int), ("padded_rows", C.c_int)]

    def as_dict(self):
        arrays = ("rows_per_bin", "ms_bin", "ms_bin_count", "t_bin", "t_bin_count", "bin_cap")
        d = {k: getattr(self, k) for k, _ in self._fields_ if k not in arrays}
        for k in arrays:
            d[k] = list(getattr(self, k))[: self.bins]
        return d


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t)
GATHERV_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p, C.POINTER(C.c_size_t), C.c_int)


class HostTransport(C.Structure):
    """bspgemm_host_transport: all-gather / gatherv callbacks on host buffers"""
    _fields_ = [("user", C.c_void_p), ("allgather", ALLGATHER_FN), ("gatherv", GATHERV_FN)]


def build():
    """Compile libbspgemm.so in-tree (hipcc --offload-arch=gfx950 + gcc)."""
    subprocess.run(["make", "-C", _ROOT, "-j8"], check=True, stdout=subprocess.DEVNULL)


_lib = None


def hip_runtimes():
    """paths of the libamdhip64 copies mapped into this process (there must be exactly one)"""
    try:
        with open("/proc/self/maps") as f:
            return sorted({ln.split()[-1] for ln in f if "libamdhip64" in ln})
    except OSError:
        return []


def check_single_hip_runtime():
    """Two HIP runtimes in one process (torch's bundled libamdhip64 + /opt/rocm's, which
    libbspgemm.so finds through its rpath when it is loaded BEFORE torch) each keep their own
    device state: handles of one are garbage to the other and the process segfaults on the first
    cross call (seen in round 1 as a crash in the RCCL stitch test).  Load order decides: whoever
    is first wins the soname.  Fail loudly instead of crashing later."""
    libs = hip_runtimes()
    if len(libs) > 1:
        raise RuntimeError("two HIP runtimes are mapped into this process: %s -- import torch BEFORE "
                           "loading libbspgemm.so (bspgemm.lib() does that itself), or do not import "
                           "torch at all" % libs)
    return libs


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise FileNotFoundError("%s is not built: run `make -C binary-spgemm_amd` "
                                "(or __graft_entry__.build())" % LIB_PATH)
    # a process that also uses torch must load torch's bundled HIP runtime first (one runtime only);
    # where torch is not installed at all the library simply uses /opt/rocm's
    import importlib.util
    if importlib.util.find_spec("torch") is not None:
        import torch  # noqa: F401  (a broken torch install raises here: loudly, not a later segfault)
    L = C.CDLL(LIB_PATH)
    check_single_hip_runtime()
    VP, PVP = C.c_void_p, C.POINTER(C.c_void_p)
    L.bspgemm_status_string.restype = C.c_char_p
    L.bspgemm_status_string.argtypes = [C.c_int]
    L.bspgemm_last_error.restype = C.c_char_p
    L.bspgemm_create.argtypes = [C.c_int, PVP]
    L.bspgemm_destroy.argtypes = [VP]
    L.bspgemm_destroy.restype = None
    L.bspgemm_set_stream.argtypes = [VP, VP]
    L.bspgemm_synchronize.argtypes = [VP]
    L.bspgemm_set_flow.argtypes = [VP, C.c_int]
    L.bspgemm_set_class_timing.argtypes = [VP, C.c_int]
    L.bspgemm_build_info.restype = C.c_char_p
    L.bspgemm_matrix_invalidate.argtypes = [VP]
    L.bspgemm_set_option.argtypes = [VP, C.c_int, C.c_int]
    L.bspgemm_get_option.argtypes = [VP, C.c_int]
    L.bspgemm_matrix_uses_blocked_table.argtypes = [VP]
    L.bspgemm_matrix_uses_padded_rows.argtypes = [VP]
    L.bspgemm_matrix_upload.argtypes = [VP, C.c_int, C.c_int, VP, VP, PVP]
    L.bspgemm_matrix_wrap_device.argtypes = [VP, C.c_int, C.c_int, C.c_int64, VP, VP, PVP]
    L.bspgemm_matrix_free.argtypes = [VP]
    L.bspgemm_matrix_free.restype = None
    L.bspgemm_matrix_rows.argtypes = [VP]
    L.bspgemm_matrix_cols.argtypes = [VP]
    L.bspgemm_matrix_nnz.argtypes = [VP]
    L.bspgemm_matrix_nnz.restype = C.c_int64
    L.bspgemm_multiply.argtypes = [VP, VP, VP, C.c_int, C.c_int, PVP]
    L.bspgemm_multiply_masked.argtypes = [VP, VP, VP, VP, C.c_int, C.c_int, PVP]
    L.bspgemm_result_rows.argtypes = [VP]
    L.bspgemm_result_nnz.argtypes = [VP]
    L.bspgemm_result_nnz.restype = C.c_int64
    L.bspgemm_result_row_ptr_device.argtypes = [VP]
    L.bspgemm_result_row_ptr_device.restype = VP
    L.bspgemm_result_col_idx_device.argtypes = [VP]
    L.bspgemm_result_col_idx_device.restype = VP
    L.bspgemm_result_download.argtypes = [VP, VP, VP, VP]
    L.bspgemm_result_free.argtypes = [VP]
    L.bspgemm_result_free.restype = None
    L.bspgemm_matrix_from_result.argtypes = [VP, VP, C.c_int, PVP]
    L.bspgemm_closure.argtypes = [VP, VP, C.c_int, PVP, C.POINTER(C.c_int)]
    L.bspgemm_row_work_prefix.argtypes = [VP, VP, VP, _I64P]
    L.bspgemm_partition_rows.argtypes = [VP, VP, VP, C.c_int, _I32P]
    L.bspgemm_last_stats.argtypes = [VP, C.POINTER(Stats)]
    L.bspgemm_stats_at.argtypes = [VP, C.c_int, C.POINTER(Stats)]
    IPP = C.POINTER(C.POINTER(C.c_int))
    L.SpGEMM_hip.argtypes = [_I32P, VP, C.c_int, _I32P, _I32P, C.c_int, IPP, _I32P, C.c_int]
    L.SpGEMM_hip_bigslice.argtypes = [_I32P, _I32P, C.c_int, _I32P, _I32P, C.c_int, IPP, _I32P,
                                      C.POINTER(C.c_int), C.c_int, C.c_int]
    L.SpGEMM_hip_mat.argtypes = [_I32P, _I32P, C.c_int, _I32P, _I32P, C.c_int, _I32P, _I32P]
    L.SpGEMM_hip_masked.argtypes = [_I32P, _I32P, C.c_int, _I32P, _I32P, C.c_int, _I32P, _I32P, IPP, _I32P,
                                    C.POINTER(C.c_int)]
    L.bspgemm_dropin_set_device.argtypes = [C.c_int]
    L.bspgemm_comm_unique_id.argtypes = [C.c_char_p]
    L.bspgemm_comm_create.argtypes = [VP, C.c_char_p, C.c_int, C.c_int, PVP]
    L.bspgemm_comm_destroy.argtypes = [VP]
    L.bspgemm_comm_destroy.restype = None
    L.bspgemm_comm_stitch_row_ptr.argtypes = [VP, VP, _I32P, PVP, VP]
    L.bspgemm_lengths_to_row_ptr.argtypes = [VP, VP, C.c_int, C.c_int, _I32P, VP, VP]
    U32PP = C.POINTER(C.POINTER(C.c_uint32))
    L.bspgemm_readCOO.argtypes = [C.c_char_p, U32PP, U32PP, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_uint32)]
    L.bspgemm_readCOO_ex.argtypes = [C.c_char_p, C.c_uint, U32PP, U32PP, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32),
                                     C.POINTER(C.c_uint32)]
    L.bspgemm_comm_create_host.argtypes = [VP, C.POINTER(HostTransport), C.c_int, C.c_int, PVP]
    L.bspgemm_comm_rank.argtypes = [VP]
    L.bspgemm_comm_size.argtypes = [VP]
    L.bspgemm_comm_gather_col_idx.argtypes = [VP, VP, _I64P, C.c_int, VP]
    L.SpGEMM_hip_multi.argtypes = [VP, _I32P, _I32P, C.c_int, _I32P, _I32P, C.c_int, IPP, _I32P, C.c_int]
    L.bspgemm_write_mtx.argtypes = [C.c_char_p, C.c_int, C.c_int, _I32P, _I32P]
    L.bspgemm_write_result_mtx.argtypes = [C.c_char_p, C.c_int, C.c_int, _I64P, _I32P]
    L.bspgemm_csr_equal.argtypes = [_I32P, _I32P, _I32P, _I32P, C.c_int]
    L.bspgemm_csr_equal64.argtypes = [_I32P, _I64P, _I32P, _I64P, C.c_int]
    L.bspgemm_gen_uniform.argtypes = [C.c_int, C.c_int, C.c_uint64, IPP, IPP]
    L.bspgemm_gen_rmat.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_double, C.c_uint64, IPP, IPP]
    L.bspgemm_gen_powerlaw.argtypes = [C.c_int, C.c_int, C.c_double, C.c_int, C.c_uint64, IPP, IPP]
    _lib = L
    return L


_libc = C.CDLL(None)
_libc.free.argtypes = [C.c_void_p]


def _chk(st, where):
    if st != 0:
        raise BspgemmError(st, where)


def _take_i32(ptr, n):
    out = np.ctypeslib.as_array(C.cast(ptr, C.POINTER(C.c_int32)), shape=(max(int(n), 1),))[: int(n)].copy()
    _libc.free(C.cast(ptr, C.c_void_p))
    return out


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


# ------------------------------------------------------------------ host utilities --------
READ_EXPAND_SYMMETRIC = 1


def readCOO(path, expand_symmetric=False):
    """bspgemm_readCOO: (row_ptr int32[M+1], col_idx int32[nnz], M, N); CSR of the transposed file.
    expand_symmetric=True: bspgemm_readCOO_ex with BSPGEMM_READ_EXPAND_SYMMETRIC (opt-in, not the
    reference's behaviour)."""
    L = lib()
    rp, ci = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
    m, n, nz = C.c_uint32(), C.c_uint32(), C.c_uint32()
    if expand_symmetric:
        st = L.bspgemm_readCOO_ex(os.fsencode(path), READ_EXPAND_SYMMETRIC, C.byref(rp), C.byref(ci), C.byref(m),
                                  C.byref(n), C.byref(nz))
    else:
        st = L.bspgemm_readCOO(os.fsencode(path), C.byref(rp), C.byref(ci), C.byref(m), C.byref(n), C.byref(nz))
    _chk(st, "readCOO(%s)" % path)
    # the CSR has one row per FILE COLUMN (N); the reference only ever reads square files (n = M)
    return _take_i32(rp, n.value + 1), _take_i32(ci, nz.value), m.value, n.value


def write_mtx(path, row_ptr, col_idx, cols=None):
    row_ptr, col_idx = _i32(row_ptr), _i32(col_idx)
    rows = row_ptr.size - 1
    _chk(lib().bspgemm_write_mtx(os.fsencode(path), rows, rows if cols is None else cols, row_ptr,
                                 col_idx if col_idx.size else np.zeros(1, np.int32)), "write_mtx")


def write_result_mtx(path, row_ptr64, col_idx, cols=None):
    row_ptr64 = np.ascontiguousarray(row_ptr64, dtype=np.int64)
    col_idx = _i32(col_idx)
    rows = row_ptr64.size - 1
    _chk(lib().bspgemm_write_result_mtx(os.fsencode(path), rows, rows if cols is None else cols, row_ptr64,
                                        col_idx if col_idx.size else np.zeros(1, np.int32)), "write_result_mtx")


def csr_equal(rp1, ci1, rp2, ci2):
    rp1 = np.ascontiguousarray(rp1, dtype=np.int64)
    rp2 = np.ascontiguousarray(rp2, dtype=np.int64)
    if rp1.size != rp2.size:
        return False
    return bool(lib().bspgemm_csr_equal64(_i32(ci1), rp1, _i32(ci2), rp2, rp1.size - 1))


def _gen(fn, n, *args):
    rp, ci = C.POINTER(C.c_int)(), C.POINTER(C.c_int)()
    _chk(fn(*args, C.byref(rp), C.byref(ci)), "generator")
    row_ptr = _take_i32(rp, n + 1)
    return row_ptr, _take_i32(ci, row_ptr[-1]), n


def gen_uniform(n, d, seed=1):
    return _gen(lib().bspgemm_gen_uniform, n, n, d, seed)


def gen_rmat(scale, edge_factor=16, abc=(0.30, 0.25, 0.25), seed=1):
    return _gen(lib().bspgemm_gen_rmat, 1 << scale, scale, edge_factor, abc[0], abc[1], abc[2], seed)


def gen_powerlaw(n, mean_degree, alpha=2.1, max_degree=0, seed=1):
    return _gen(lib().bspgemm_gen_powerlaw, n, n, mean_degree, alpha, max_degree, seed)


# ------------------------------------------------------------------ native handle API -----
class Context:
    """bspgemm_context: one GPU, one stream, reusable workspaces."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        L = lib()
        check_single_hip_runtime()
        _chk(L.bspgemm_create(device, C.byref(self._h)), "bspgemm_create")
        self.device = device
        self._children = weakref.WeakSet()      # matrices/results must go before their context

    def close(self):
        if self._h:
            for ch in list(self._children):
                ch.free()
            lib().bspgemm_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, hip_stream_ptr):
        _chk(lib().bspgemm_set_stream(self._h, C.c_void_p(hip_stream_ptr)), "set_stream")

    def synchronize(self):
        _chk(lib().bspgemm_synchronize(self._h), "synchronize")

    def set_flow(self, flow):
        """"auto" | "upper-bound" | "exact" (BSPGEMM_FLOW_*, include/bspgemm.h)"""
        _chk(lib().bspgemm_set_flow(self._h, FLOWS[flow]), "set_flow")

    def set_option(self, name, value):
        """bspgemm_set_option: "class_streams" 1..3, "blocked_extents" -1/0/1, "check" 0/1, "small_path" -1/0/1, "padded_rows" -1/0/1"""
        _chk(lib().bspgemm_set_option(self._h, OPTIONS[name], int(value)), "set_option(%s)" % name)

    def get_option(self, name):
        return lib().bspgemm_get_option(self._h, OPTIONS[name])

    def set_class_timing(self, on):
        """event brackets around every class launch (stats: ms_bin, t_bin, ...); off by default: they cost ~1 %"""
        _chk(lib().bspgemm_set_class_timing(self._h, 1 if on else 0), "set_class_timing")

    def upload(self, row_ptr, col_idx, cols, row0=0, rows=None):
        """Host CSR -> device.  row0/rows select an interior row range (absolute row_ptr values)."""
        row_ptr, col_idx = _i32(row_ptr), _i32(col_idx)
        rows = row_ptr.size - 1 - row0 if rows is None else rows
        m = C.c_void_p()
        _chk(lib().bspgemm_matrix_upload(self._h, rows, cols, C.c_void_p(row_ptr.ctypes.data + 4 * row0),
                                         C.c_void_p(col_idx.ctypes.data), C.byref(m)), "matrix_upload")
        return Matrix(self, m, keep=None)

    def wrap_device(self, rows, cols, nnz, d_row_ptr, d_col_idx, keep=None):
        """Adopt device arrays (e.g. torch tensors' data_ptr()); `keep` holds their owners alive."""
        m = C.c_void_p()
        _chk(lib().bspgemm_matrix_wrap_device(self._h, rows, cols, nnz, C.c_void_p(d_row_ptr),
                                              C.c_void_p(d_col_idx), C.byref(m)), "matrix_wrap_device")
        return Matrix(self, m, keep=keep)

    def multiply(self, A, B, row_begin=0, row_end=None):
        row_end = A.rows if row_end is None else row_end
        r = C.c_void_p()
        _chk(lib().bspgemm_multiply(self._h, A._h, B._h, row_begin, row_end, C.byref(r)), "bspgemm_multiply")
        return Result(self, r)

    def multiply_masked(self, A, B, F, row_begin=0, row_end=None):
        row_end = A.rows if row_end is None else row_end
        r = C.c_void_p()
        _chk(lib().bspgemm_multiply_masked(self._h, A._h, B._h, F._h, row_begin, row_end, C.byref(r)),
             "bspgemm_multiply_masked")
        return Result(self, r)

    def matrix_from_result(self, result, cols):
        m = C.c_void_p()
        _chk(lib().bspgemm_matrix_from_result(self._h, result._h, cols, C.byref(m)), "matrix_from_result")
        return Matrix(self, m, keep=None)

    def closure(self, A, max_iter=64):
        """reflexive-transitive closure by repeated squaring; returns (Result, products computed)"""
        r, it = C.c_void_p(), C.c_int()
        _chk(lib().bspgemm_closure(self._h, A._h, max_iter, C.byref(r), C.byref(it)), "bspgemm_closure")
        return Result(self, r), it.value

    def lengths_to_row_ptr(self, d_lengths, world, width, bounds, d_row_ptr, hip_stream=None):
        """gathered int32 row lengths (device) -> global int64 row_ptr (device); see bspgemm.h"""
        b = np.ascontiguousarray(bounds, dtype=np.int32)
        _chk(lib().bspgemm_lengths_to_row_ptr(self._h, C.c_void_p(int(d_lengths)), int(world), int(width), b,
                                              C.c_void_p(int(d_row_ptr)), C.c_void_p(hip_stream or 0)), "lengths_to_row_ptr")

    def stats(self, age=0):
        """counters and HIP-event times of the last multiply (age 0) or of an earlier one (age <= 15)"""
        s = Stats()
        _chk(lib().bspgemm_stats_at(self._h, age, C.byref(s)), "stats_at")
        return s.as_dict()

    def row_work_prefix(self, A, B):
        out = np.zeros(A.rows + 1, dtype=np.int64)
        _chk(lib().bspgemm_row_work_prefix(self._h, A._h, B._h, out), "row_work_prefix")
        return out

    def partition_rows(self, A, B, parts):
        out = np.zeros(parts + 1, dtype=np.int32)
        _chk(lib().bspgemm_partition_rows(self._h, A._h, B._h, parts, out), "partition_rows")
        return out


class Matrix:
    def __init__(self, ctx, handle, keep):
        self.ctx, self._h, self._keep = ctx, handle, keep
        ctx._children.add(self)
        self.rows = lib().bspgemm_matrix_rows(handle)
        self.cols = lib().bspgemm_matrix_cols(handle)
        self.nnz = lib().bspgemm_matrix_nnz(handle)

    def invalidate(self):
        """bspgemm_matrix_invalidate: the wrapped device arrays were rewritten in place; derived tables are rebuilt"""
        _chk(lib().bspgemm_matrix_invalidate(self._h), "matrix_invalidate")

    @property
    def uses_blocked_table(self):
        """1 / 0 once the operand has been used as B (which prepass kernel it gets), -1 before"""
        return lib().bspgemm_matrix_uses_blocked_table(self._h)

    @property
    def uses_padded_rows(self):
        """1 / 0 once the operand has been used as B (whether its rows are gathered from the padded copy), -1 before"""
        return lib().bspgemm_matrix_uses_padded_rows(self._h)

    def free(self):
        if self._h:
            lib().bspgemm_matrix_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Result:
    def __init__(self, ctx, handle):
        self.ctx, self._h = ctx, handle
        ctx._children.add(self)
        self.rows = lib().bspgemm_result_rows(handle)
        self.nnz = lib().bspgemm_result_nnz(handle)

    @property
    def row_ptr_device(self):
        return lib().bspgemm_result_row_ptr_device(self._h)

    @property
    def col_idx_device(self):
        return lib().bspgemm_result_col_idx_device(self._h)

    def download(self, col_idx=True):
        rp = np.zeros(self.rows + 1, dtype=np.int64)
        ci = np.zeros(self.nnz if col_idx else 0, dtype=np.int32)
        _chk(lib().bspgemm_result_download(self.ctx._h, self._h, C.c_void_p(rp.ctypes.data),
                                           C.c_void_p(ci.ctypes.data) if (col_idx and self.nnz) else None),
             "result_download")
        return rp, ci

    def free(self):
        if self._h:
            lib().bspgemm_result_free(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


# ------------------------------------------------------------------ int32 drop-ins --------
def SpGEMM_hip(Acol, Arow, An, Bcol, Brow, Bm, tBlock=0, row0=0):
    """SpGEMM_omp-shaped call (final/SpGEMM_mpi_omp.c:71-74).  Returns (Crow int32[An+1], Ccol)."""
    Acol, Arow, Bcol, Brow = map(_i32, (Acol, Arow, Bcol, Brow))
    crow = np.zeros(An + 1, dtype=np.int32)
    cc = C.POINTER(C.c_int)()
    st = lib().SpGEMM_hip(Acol, C.c_void_p(Arow.ctypes.data + 4 * row0), An, Bcol, Brow, Bm, C.byref(cc), crow, tBlock)
    _chk(st, "SpGEMM_hip")
    return crow, _take_i32(cc, crow[-1])


def SpGEMM_hip_bigslice(Acol, Arow, An, Bcol, Brow, Bm, start_row, end_row):
    Acol, Arow, Bcol, Brow = map(_i32, (Acol, Arow, Bcol, Brow))
    crow = np.zeros(end_row - start_row + 1, dtype=np.int32)
    csize = C.c_int(max(Bm, 1))
    _libc.malloc.restype = C.c_void_p
    _libc.malloc.argtypes = [C.c_size_t]
    cc = C.cast(_libc.malloc(csize.value * 4), C.POINTER(C.c_int))
    st = lib().SpGEMM_hip_bigslice(Acol, Arow, An, Bcol, Brow, Bm, C.byref(cc), crow, C.byref(csize), start_row, end_row)
    _chk(st, "SpGEMM_hip_bigslice")
    return crow, _take_i32(cc, crow[-1])


def SpGEMM_hip_mat(Acol, Arow, An, Bcol, Brow, Bm, nnz_c):
    Acol, Arow, Bcol, Brow = map(_i32, (Acol, Arow, Bcol, Brow))
    crow = np.zeros(An + 1, dtype=np.int32)
    ccol = np.zeros(max(int(nnz_c), 1), dtype=np.int32)
    _chk(lib().SpGEMM_hip_mat(Acol, Arow, An, Bcol, Brow, Bm, ccol, crow), "SpGEMM_hip_mat")
    return crow, ccol[: int(nnz_c)]


def SpGEMM_hip_masked(Acol, Arow, An, Bcol, Brow, Bm, Fcol, Frow):
    """SpGEMM_masked-shaped call (final/SpGEMM_mpi_omp.c:232-235): C = F .* (A*B)."""
    Acol, Arow, Bcol, Brow, Fcol, Frow = map(_i32, (Acol, Arow, Bcol, Brow, Fcol, Frow))
    crow = np.zeros(An + 1, dtype=np.int32)
    csize = C.c_int(max(An, 1))
    _libc.malloc.restype = C.c_void_p
    _libc.malloc.argtypes = [C.c_size_t]
    cc = C.cast(_libc.malloc(csize.value * 4), C.POINTER(C.c_int))
    st = lib().SpGEMM_hip_masked(Acol, Arow, An, Bcol, Brow, Bm, Fcol, Frow, C.byref(cc), crow, C.byref(csize))
    _chk(st, "SpGEMM_hip_masked")
    return crow, _take_i32(cc, crow[-1])
