"""CPU-only tests of the product's host C side (loader, writer, comparator, generators)."""
import os

import numpy as np
import pytest

import bspgemm
import gen
from oracle import oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_readcoo_matches_reference_goldens():
    g = np.load(os.path.join(GOLDEN, "validity.npz"), allow_pickle=False)
    rp, ci, m, n = bspgemm.readCOO(os.path.join(GOLDEN, "validity_test.mtx"))
    assert (m, n) == (50000, 50000)
    assert np.array_equal(rp, g["a_rp"]) and np.array_equal(ci, g["a_ci"])
    g = np.load(os.path.join(GOLDEN, "loader_case_readcoo.npz"), allow_pickle=False)
    rp, ci, m, n = bspgemm.readCOO(os.path.join(GOLDEN, "loader_case.mtx"))
    assert np.array_equal(rp, g["row_ptr"]) and np.array_equal(ci, g["col_idx"])


def test_readcoo_error_behaviour(tmp_path):
    """reference: fopen failure -> exit(1) (utils.c:54), bad banner -> message + exit(1) (:56-59),
    bad size line -> exit(1) WITHOUT a message (:60-61): a status of its own, so the CLI can tell"""
    with pytest.raises(bspgemm.BspgemmError) as e:
        bspgemm.readCOO(str(tmp_path / "nope.mtx"))
    assert e.value.status == 6
    p = tmp_path / "bad.mtx"
    p.write_text("%MatrixMarket matrix coordinate pattern general\n2 2 1\n1 1\n")
    with pytest.raises(bspgemm.BspgemmError) as e:
        bspgemm.readCOO(str(p))
    assert e.value.status == 7
    p.write_text("%%MatrixMarket matrix coordinate pattern general\n% only comments\n")
    with pytest.raises(bspgemm.BspgemmError) as e:
        bspgemm.readCOO(str(p))
    assert e.value.status == 9
    p.write_text("%%MatrixMarket matrix coordinate pattern general\n2 2 x\n")
    with pytest.raises(bspgemm.BspgemmError) as e:
        bspgemm.readCOO(str(p))
    assert e.value.status == 9


def test_readcoo_symmetric_expansion_is_opt_in(tmp_path):
    """f2: the reference parses the symmetry token (mmio.c:96-179) and ignores it (utils.c:66-71);
    the default stays reference-exact, BSPGEMM_READ_EXPAND_SYMMETRIC mirrors the stored triangle"""
    p = tmp_path / "s.mtx"
    p.write_text("%%MatrixMarket matrix coordinate pattern symmetric\n4 4 4\n1 1\n3 1\n4 2\n4 4\n")
    rp, ci, m, n = bspgemm.readCOO(str(p))
    orp, oci, _, _ = O.read_mtx(str(p))                       # oracle = reference semantics: not expanded
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci) and rp[-1] == 4
    rp2, ci2, m2, n2 = bspgemm.readCOO(str(p), expand_symmetric=True)
    assert (m2, n2) == (4, 4) and rp2[-1] == 6
    import scipy.sparse as sp
    full = sp.csr_matrix((np.ones(6), ci2, rp2), shape=(4, 4)).toarray() > 0
    want = np.zeros((4, 4), bool)
    for i, j in ((0, 0), (2, 0), (3, 1), (3, 3)):
        want[i, j] = want[j, i] = True
    assert np.array_equal(full, want)                           # symmetric: the transpose quirk is invisible
    # a general file is untouched by the flag; a real symmetric file has its values skipped
    g = tmp_path / "g.mtx"
    g.write_text("%%MatrixMarket matrix coordinate pattern general\n3 3 2\n2 1\n3 2\n")
    a = bspgemm.readCOO(str(g))
    b = bspgemm.readCOO(str(g), expand_symmetric=True)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    r = tmp_path / "r.mtx"
    r.write_text("%%MatrixMarket matrix coordinate real symmetric\n3 3 2\n2 1 0.5\n3 3 2.0\n")
    rp3, ci3, _, _ = bspgemm.readCOO(str(r), expand_symmetric=True)
    assert rp3.tolist() == [0, 1, 2, 3] and ci3.tolist() == [1, 0, 2]


def test_readcoo_transposes_and_keeps_file_order(tmp_path):
    p = tmp_path / "t.mtx"
    # file entries (row, col): (1,2) (3,2) (2,2) (1,1) (1,2)dup   -> CSR row = file col, stable
    p.write_text("%%MatrixMarket matrix coordinate pattern general\n3 3 5\n1 2\n3 2\n2 2\n1 1\n1 2\n")
    rp, ci, m, n = bspgemm.readCOO(str(p))
    assert rp.tolist() == [0, 1, 5, 5]
    assert ci.tolist() == [0, 0, 2, 1, 0]
    orp, oci, _, _ = O.read_mtx(str(p))
    assert np.array_equal(rp, orp) and np.array_equal(ci, oci)


def test_readcoo_skips_values_of_real_files(tmp_path):
    p = tmp_path / "r.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n2 2 2\n1 2 3.5\n2 1 -1e3\n")
    rp, ci, m, n = bspgemm.readCOO(str(p))
    assert rp.tolist() == [0, 1, 2] and ci.tolist() == [1, 0]


def test_write_then_read_roundtrip(tmp_path):
    rp, ci, n = gen.uniform(300, 5, seed=3)
    p = str(tmp_path / "w.mtx")
    bspgemm.write_mtx(p, rp, ci)
    rp2, ci2, m, nn = bspgemm.readCOO(p)
    assert (m, nn) == (n, n) and np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
    # and the reference's own loader agrees, where it is available
    R = O.reference()
    if R is not None:
        rp3, ci3, _, _ = R.read_mtx(p)
        assert np.array_equal(rp, rp3) and np.array_equal(ci, ci3)
    crow, ccol = O.spgemm(rp, ci, rp, ci, n)
    bspgemm.write_result_mtx(p, crow, ccol)
    rp4, ci4, _, _ = bspgemm.readCOO(p)
    assert np.array_equal(rp4, crow) and np.array_equal(ci4, ccol)


def test_rectangular_write_read_roundtrip(tmp_path):
    """a CSR with R rows and K columns is the file matrix K x R (transposing loader, utils.c:77)"""
    rp, ci = gen.uniform_rect(37, 91, 4, seed=9)
    p = str(tmp_path / "rect.mtx")
    bspgemm.write_mtx(p, rp, ci, cols=91)
    assert open(p).read().split("\n")[1].split()[:2] == ["91", "37"]
    rp2, ci2, m, n = bspgemm.readCOO(p)
    assert (m, n) == (91, 37) and np.array_equal(rp, rp2) and np.array_equal(ci, ci2)


def test_csr_equal_is_spgemm_valid():
    rp, ci, n = gen.uniform(100, 4, seed=5)
    assert bspgemm.csr_equal(rp, ci, rp, ci)
    ci2 = ci.copy()
    ci2[-1] ^= 1
    assert not bspgemm.csr_equal(rp, ci, rp, ci2)
    rp2 = rp.copy()
    rp2[50] += 1
    assert not bspgemm.csr_equal(rp, ci, rp2, ci)


@pytest.mark.parametrize("make", [
    lambda s: bspgemm.gen_uniform(5000, 16, seed=s),
    lambda s: bspgemm.gen_rmat(12, 16, (0.30, 0.25, 0.25), seed=s),
    lambda s: bspgemm.gen_rmat(11, 8, (0.57, 0.19, 0.19), seed=s),
    lambda s: bspgemm.gen_powerlaw(4096, 32, seed=s),
])
def test_generators_wellformed_and_deterministic(make):
    rp, ci, n = make(1)
    assert rp[0] == 0 and rp[-1] == ci.size and np.all(np.diff(rp) >= 0)
    assert ci.min() >= 0 and ci.max() < n
    row_of = np.repeat(np.arange(n), np.diff(rp))
    d = np.diff(ci.astype(np.int64))
    same_row = row_of[1:] == row_of[:-1]
    assert np.all(d[same_row] > 0), "rows must be strictly ascending (sorted, no duplicates)"
    rp2, ci2, _ = make(1)
    assert np.array_equal(rp, rp2) and np.array_equal(ci, ci2)
    rp3, ci3, _ = make(2)
    assert not (np.array_equal(rp, rp3) and np.array_equal(ci, ci3))


def test_generator_shapes():
    rp, ci, n = bspgemm.gen_uniform(1 << 14, 16, seed=1)
    assert n == 1 << 14 and 15.9 * n < rp[-1] <= 16 * n
    rp, ci, n = bspgemm.gen_rmat(14, 16, (0.30, 0.25, 0.25), seed=1)
    assert n == 1 << 14 and 0.9 * 16 * n < rp[-1] <= 16 * n
    rp, ci, n = bspgemm.gen_powerlaw(1 << 14, 64, seed=1)
    deg = np.diff(rp)
    # draws are mean 64 per row; skewed column choice collapses many duplicates at small n
    assert deg.min() >= 1 and deg.max() <= n // 16 and 20 * n < rp[-1] <= 64 * n
    assert deg.max() > 20 * np.median(deg), "expected a heavy tail"


def test_host_c_under_address_and_ub_sanitizers(tmp_path):
    """the reference ships no sanitizer build.  oracle/Makefile builds the PRODUCT's host C (loader,
    writers, generators, comparators, the drop-ins' host helpers) together with the restatement under
    -fsanitize=address,undefined (asan/host_asan, CPU only) and this walks every entry point, including the loader's
    rejection paths on ten malformed files; any report aborts the program."""
    import subprocess
    exe = os.path.join(ROOT, "oracle", "asan", "host_asan")
    if not os.path.exists(exe):
        subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], check=True, stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1", OMP_NUM_THREADS="4")
    r = subprocess.run([exe, str(tmp_path), os.path.join(ROOT, "tests", "golden", "validity_test.mtx")],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "host_asan ok" in r.stdout and "ERROR" not in r.stderr and "runtime error" not in r.stderr, r.stderr


def test_powerlaw_generator_hits_the_requested_mean_degree():
    """the generator drew deg_i columns per row ONCE and let duplicates collapse (mean 52 for a requested
    64 at BASELINE config 5); rows are now topped up to deg_i DISTINCT columns"""
    for n, d in ((1 << 14, 64), (20000, 24), (3000, 7)):
        rp, ci, _ = bspgemm.gen_powerlaw(n, d, seed=3)
        deg = np.diff(rp)
        assert rp[-1] == n * d, (n, d, rp[-1] / n)
        assert deg.min() >= 1 and deg.max() <= max(n // 16, 1)
        for r in (0, n // 2, int(np.argmax(deg))):
            row = ci[rp[r]:rp[r + 1]]
            assert np.all(np.diff(row) > 0)                  # sorted, duplicate-free
