"""Pins the CPU restatement (oracle/bspgemm_oracle.c) to the reference.

(1) against the committed golden vectors in tests/golden/ -- outputs of the reference itself
    (oracle/_ref, built from /root/reference by oracle/Makefile; script: make_golden.py);
(2) against the compiled reference live, when oracle/_ref is present, on fresh seeded inputs.
CPU only; no GPU needed.
"""
import glob
import os

import numpy as np
import pytest

import gen
from oracle import oracle as O

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


def test_validity_fixture_golden_facts():
    """n=50000, nnz(A)=25000, nnz(A*A)=12502, max|A_i|=6, max|C_i|=7,
    8938 non-empty C rows."""
    rp, ci, m, n = O.read_mtx(os.path.join(GOLDEN, "validity_test.mtx"))
    g = _load("validity")
    assert (m, n, rp[-1]) == (50000, 50000, 25000)
    assert np.array_equal(rp, g["a_rp"]) and np.array_equal(ci, g["a_ci"])     # loader == readCOO
    crow, ccol = O.spgemm(rp, ci, rp, ci, m)
    assert crow[-1] == 12502
    assert np.array_equal(crow, g["c_rp"]) and np.array_equal(ccol, g["c_ci"])
    assert np.diff(rp).max() == 6 and np.diff(crow).max() == 7
    assert int((np.diff(crow) > 0).sum()) == 8938
    assert O.count_products(rp, ci, rp) == 12502


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*_bigslice.npz"))),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_bigslice_goldens(path):
    g = np.load(path, allow_pickle=False)
    n = int(g["n"])
    crow, ccol = O.spgemm(g["a_rp"], g["a_ci"], g["a_rp"], g["a_ci"], n)
    assert np.array_equal(crow, g["c_rp"])
    assert np.array_equal(ccol, g["c_ci"])
    assert O.csr_equal(crow, ccol, g["c_rp"], g["c_ci"])
    # the restated OpenMP stitching must give the same matrix for any slice size
    for tb in (1, 7, 64, n if n else 1):
        r2, c2 = O.spgemm_omp(g["a_rp"], g["a_ci"], g["a_rp"], g["a_ci"], n, tb, 2)
        assert np.array_equal(r2, g["c_rp"]) and np.array_equal(c2, g["c_ci"])


def test_omp_interior_rows_golden():
    g = _load("uniform_n4096_rows1024_3072_omp")
    crow, ccol = O.spgemm_omp(g["a_rp"], g["a_ci"], g["a_rp"], g["a_ci"], int(g["n"]), int(g["tblock"]), 3,
                              row0=int(g["row0"]), rows=int(g["rows"]))
    assert np.array_equal(crow, g["c_rp"]) and np.array_equal(ccol, g["c_ci"])
    r2, c2 = O.spgemm_rows(g["a_rp"], g["a_ci"], g["a_rp"], g["a_ci"], int(g["n"]),
                           int(g["row0"]), int(g["row0"]) + int(g["rows"]))
    assert np.array_equal(r2, g["c_rp"]) and np.array_equal(c2, g["c_ci"])


def test_rect_a_ne_b_golden():
    g = _load("rect_300x200x250_mat")
    crow, ccol = O.spgemm(g["a_rp"], g["a_ci"], g["b_rp"], g["b_ci"], int(g["bm"]))
    assert np.array_equal(crow, g["c_rp"]) and np.array_equal(ccol, g["c_ci"])


def test_masked_golden():
    g = _load("masked_n512")
    crow, ccol = O.spgemm_masked(g["a_rp"], g["a_ci"], g["a_rp"], g["a_ci"], int(g["n"]), g["f_rp"], g["f_ci"])
    assert np.array_equal(crow, g["c_rp"]) and np.array_equal(ccol, g["c_ci"])


def test_loader_golden():
    g = _load("loader_case_readcoo")
    rp, ci, m, n = O.read_mtx(os.path.join(GOLDEN, "loader_case.mtx"))
    assert (m, n) == (int(g["m"]), int(g["n"]))
    assert np.array_equal(rp, g["row_ptr"]) and np.array_equal(ci, g["col_idx"])


def test_loader_rejects_bad_banner(tmp_path):
    p = tmp_path / "bad.mtx"
    p.write_text("%MatrixMarket matrix coordinate pattern general\n2 2 1\n1 1\n")
    with pytest.raises(OSError):
        O.read_mtx(str(p))
    with pytest.raises(OSError):
        O.read_mtx(str(tmp_path / "missing.mtx"))


def test_scipy_second_opinion():
    """scipy stands in for Matlab's A*B>0 (Matlab/test_SpGEMM.m:20) -- A != B, random."""
    sp = pytest.importorskip("scipy.sparse")
    a_rp, a_ci = gen.uniform_rect(700, 900, 6, seed=21)
    b_rp, b_ci = gen.uniform_rect(900, 1100, 4, seed=22)
    A = sp.csr_matrix((np.ones(a_ci.size, np.int8), a_ci, a_rp), shape=(700, 900))
    B = sp.csr_matrix((np.ones(b_ci.size, np.int8), b_ci, b_rp), shape=(900, 1100))
    Cs = (A.astype(np.int32) @ B.astype(np.int32)).tocsr()
    Cs.sort_indices()
    crow, ccol = O.spgemm(a_rp, a_ci, b_rp, b_ci, 1100)
    assert np.array_equal(crow, Cs.indptr) and np.array_equal(ccol, Cs.indices)


# ---- live comparison with the compiled reference (this container / prebuilt on the GPU box) ----
needs_ref = pytest.mark.skipif(O.reference() is None, reason="oracle/_ref not built")


@needs_ref
@pytest.mark.parametrize("maker,args", [
    (gen.uniform, (3000, 10, 101)),
    (gen.rmat, (12, 8, (0.57, 0.19, 0.19, 0.05), 102)),
    (gen.with_special_rows, (777, 5, 103)),
    (gen.dups_unsorted, (640, 12, 104)),
    (gen.powerlaw, (2048, 12, 105)),
])
def test_live_reference_bigslice(maker, args):
    R = O.reference()
    rp, ci, n = maker(*args)
    crow, ccol = O.spgemm(rp, ci, rp, ci, n)
    rrow, rcol = R.bigslice(rp, ci, rp, ci, n, 0, n)
    assert np.array_equal(crow, rrow) and np.array_equal(ccol, rcol)


@needs_ref
def test_live_reference_omp_and_loader(tmp_path):
    R = O.reference()
    rp, ci, n = gen.uniform(2048, 7, 106)
    rrow, rcol = R.omp(rp, ci, rp, ci, n, 128, row0=512, rows=1024)
    crow, ccol = O.spgemm_omp(rp, ci, rp, ci, n, 128, 2, row0=512, rows=1024)
    assert np.array_equal(crow, rrow) and np.array_equal(ccol, rcol)
    a, b, m, nn = R.read_mtx(os.path.join(GOLDEN, "validity_test.mtx"))
    c, d, m2, nn2 = O.read_mtx(os.path.join(GOLDEN, "validity_test.mtx"))
    assert (m, nn) == (m2, nn2) and np.array_equal(a, c) and np.array_equal(b, d)
