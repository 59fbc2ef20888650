#!/usr/bin/env python3
"""Mint the golden vectors in tests/golden/ by RUNNING THE REFERENCE (oracle/_ref, compiled
from /root/reference by oracle/Makefile) on small seeded inputs.  Run in the build container:

    make -C oracle && python tests/golden/make_golden.py

Every .npz holds plain integer arrays only (inputs and the reference's outputs); they load
with numpy.load(allow_pickle=False).  validity_test.mtx is the reference's own committed test
input (Matlab/validity_test.mtx), copied as a data fixture.

Which reference entry point produced which golden:
  *_bigslice  : SpGEMM_bigslice   final/SpGEMM_mpi_omp.c:15-58   (serial, whole matrix)
  *_omp       : SpGEMM_omp        final/SpGEMM_mpi_omp.c:71-143  (slices, interior Arow pointer)
  *_mat       : SpGEMM_mat        Matlab/inc/BSpGEMM.c:9-47      (A != B)
  *_masked    : SpGEMM_masked     final/SpGEMM_mpi_omp.c:232-288
  *_readcoo   : readCOO           final/utils.c:47-81            (loader, transposing)
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import oracle as O  # noqa: E402
import gen  # noqa: E402  (tests/gen.py: seeded numpy generators)


def save(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: np.asarray(v) for k, v in arrays.items()})
    print("%-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


def main():
    R = O.reference()
    if R is None:
        sys.exit("oracle/_ref is not built: run `make -C oracle` where /root/reference exists")

    # 1. the reference's own committed input (make test, final/Makefile:11-12)
    src = "/root/reference/Matlab/validity_test.mtx"
    dst = os.path.join(HERE, "validity_test.mtx")
    shutil.copyfile(src, dst)
    rp, ci, m, n = R.read_mtx(dst)
    crow, ccol = R.bigslice(rp, ci, rp, ci, m, 0, m)
    orow, ocol = R.omp(rp, ci, rp, ci, m, 6250)          # the make-test slice size
    assert (orow == crow).all() and (ocol == ccol).all()
    save("validity", a_rp=rp, a_ci=ci, m=m, n=n, c_rp=crow, c_ci=ccol)

    # 2..6  A*A on seeded generated matrices through SpGEMM_bigslice
    cases = {
        "uniform_n4096_d8": gen.uniform(4096, 8, seed=1),
        "rmat_s10_e8_g500": gen.rmat(10, 8, (0.57, 0.19, 0.19, 0.05), seed=2),
        "rmat_s11_e16_mild": gen.rmat(11, 16, (0.30, 0.25, 0.25, 0.20), seed=3),
        "emptyrows_fullrow_n1000": gen.with_special_rows(1000, 6, seed=4),
        "dups_unsorted_n513": gen.dups_unsorted(513, 9, seed=5),
        "tiny_n1": (np.array([0, 1], np.int32), np.array([0], np.int32), 1),
        "empty_n64": (np.zeros(65, np.int32), np.zeros(0, np.int32), 64),
        "banded_n2048": gen.banded(2048, 5, seed=6),
    }
    for name, (rp, ci, n) in cases.items():
        crow, ccol = R.bigslice(rp, ci, rp, ci, n, 0, n)
        save(name + "_bigslice", a_rp=rp, a_ci=ci, n=n, c_rp=crow, c_ci=ccol)

    # 7. SpGEMM_omp on an interior row range (what SpGEMM_mpi hands each rank, :171)
    rp, ci, n = cases["uniform_n4096_d8"]
    orow, ocol = R.omp(rp, ci, rp, ci, n, 64, row0=1024, rows=2048)
    save("uniform_n4096_rows1024_3072_omp", a_rp=rp, a_ci=ci, n=n, row0=1024, rows=2048, tblock=64,
         c_rp=orow, c_ci=ocol)

    # 8. A != B, rectangular, through the Matlab twin SpGEMM_mat (needs Bm <= An)
    a_rp, a_ci = gen.uniform_rect(300, 200, 5, seed=7)
    b_rp, b_ci = gen.uniform_rect(200, 250, 7, seed=8)
    orp, oci = O.spgemm(a_rp, a_ci, b_rp, b_ci, 250)      # only to learn nnz for the preallocation
    crow, ccol = R.spgemm_mat(a_rp, a_ci, b_rp, b_ci, 250, orp[-1])
    save("rect_300x200x250_mat", a_rp=a_rp, a_ci=a_ci, b_rp=b_rp, b_ci=b_ci, an=300, bm=250,
         c_rp=crow, c_ci=ccol)

    # 9. masked product
    rp, ci, n = gen.uniform(512, 6, seed=9)
    f_rp, f_ci, _ = gen.uniform(512, 40, seed=10)
    crow, ccol = R.masked(rp, ci, rp, ci, n, f_rp, f_ci)
    save("masked_n512", a_rp=rp, a_ci=ci, f_rp=f_rp, f_ci=f_ci, n=n, c_rp=crow, c_ci=ccol)

    # 10. loader: small .mtx files written here, parsed by the reference's readCOO
    mtx = os.path.join(HERE, "loader_case.mtx")
    rng = np.random.default_rng(11)
    ent = rng.integers(1, 41, size=(150, 2))
    with open(mtx, "w") as f:
        f.write("%%MatrixMarket matrix coordinate pattern general\n% a comment\n%another\n40 40 150\n")
        for i, j in ent:
            f.write("%d %d\n" % (i, j))
    rp, ci, m, n = R.read_mtx(mtx)
    save("loader_case_readcoo", row_ptr=rp, col_idx=ci, m=m, n=n)


if __name__ == "__main__":
    main()
