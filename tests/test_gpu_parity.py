"""GPU parity: the HIP path, called through the C ABI (libbspgemm.so), must be BIT-EXACT
(identical row_ptr, identical sorted col_idx) against
  (1) the committed golden vectors produced by the reference itself (tests/golden/),
  (2) the CPU oracle (oracle/bspgemm_oracle.c, pinned to the reference) on seeded inputs that
      exercise every accumulator path: capacity classes 64..2048, LEVELS 1..4 of the rank bitmap,
      the dense-window kernel (one and several windows), empty / full / duplicate rows,
      rectangular A != B, interior row ranges,
  (3) size-independent properties at BASELINE sizes (sortedness, row_ptr monotone, shard
      concatenation == whole product, idempotence of the boolean square on a closure).
Integer work: the bar is exact equality; there is no tolerance anywhere in this file.
"""
import glob
import os

import numpy as np
import pytest

import bspgemm
import gen
from oracle import oracle as O

pytestmark = pytest.mark.gpu
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


FLOW_ID = {"upper-bound": 1, "exact": 2}        # BSPGEMM_FLOW_* as bspgemm_stats.flow reports it


@pytest.fixture(scope="module", params=["upper-bound", "exact"])
def ctx(request):
    """every test of this file that takes `ctx` runs under both ways from row sizes to C.col_idx (include/bspgemm.h,
    BSPGEMM_FLOW_*): upper-bound placement + compaction, and exact symbolic sizes + emit in place"""
    c = bspgemm.Context(0)
    c.set_flow(request.param)
    c.flow_name = request.param
    yield c
    c.close()


@pytest.fixture(scope="module")
def mctx():
    """the masked product has ONE flow (bspgemm_multiply_masked always places rows by mask length and compacts):
    its tests are not repeated per flow"""
    c = bspgemm.Context(0)
    yield c
    c.close()


def hip_product(ctx, a_rp, a_ci, a_cols, b_rp, b_ci, b_cols, r0=0, r1=None):
    A = ctx.upload(a_rp, a_ci, a_cols)
    B = A if (b_rp is a_rp and b_ci is a_ci) else ctx.upload(b_rp, b_ci, b_cols)
    C = ctx.multiply(A, B, r0, A.rows if r1 is None else r1)
    rp, ci = C.download()
    st = ctx.stats()
    C.free()
    want = FLOW_ID.get(getattr(ctx, "flow_name", None))
    assert want is None or st["flow"] == want, "asked for the %s flow, bspgemm_stats says %d ran" % (ctx.flow_name, st["flow"])
    return rp, ci, st


def assert_same(rp, ci, erp, eci):
    assert rp.shape == np.asarray(erp).shape
    assert np.array_equal(rp, erp), "row_ptr differs (first at %s)" % np.flatnonzero(rp != erp)[:5]
    assert np.array_equal(ci, eci), "col_idx differs (first at %s)" % np.flatnonzero(ci != eci)[:5]


# ---------------------------------------------------------------- (1) golden vectors ------
def test_validity_fixture(ctx):
    """BASELINE config 1: Matlab/validity_test.mtx A*A, nnz 12502 -- loader + product."""
    rp, ci, m, n = bspgemm.readCOO(os.path.join(GOLDEN, "validity_test.mtx"))
    g = np.load(os.path.join(GOLDEN, "validity.npz"), allow_pickle=False)
    crp, cci, st = hip_product(ctx, rp, ci, n, rp, ci, n)
    assert crp[-1] == 12502 and st["nnz_c"] == 12502 and st["products"] == 12502
    assert_same(crp, cci, g["c_rp"], g["c_ci"])


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(GOLDEN, "*_bigslice.npz"))),
                         ids=lambda p: os.path.basename(p)[:-4])
def test_bigslice_goldens(ctx, path):
    g = np.load(path, allow_pickle=False)
    n = int(g["n"])
    a_rp, a_ci = g["a_rp"], g["a_ci"]
    crp, cci, _ = hip_product(ctx, a_rp, a_ci, n, a_rp, a_ci, n)
    assert_same(crp, cci, g["c_rp"], g["c_ci"])


def test_omp_interior_rows_golden(ctx):
    g = np.load(os.path.join(GOLDEN, "uniform_n4096_rows1024_3072_omp.npz"), allow_pickle=False)
    n, r0, rows = int(g["n"]), int(g["row0"]), int(g["rows"])
    crp, cci, _ = hip_product(ctx, g["a_rp"], g["a_ci"], n, g["a_rp"], g["a_ci"], n, r0, r0 + rows)
    assert_same(crp, cci, g["c_rp"], g["c_ci"])
    # the int32 drop-in with an interior Arow pointer, exactly how SpGEMM_mpi calls SpGEMM_omp (:171)
    crow, ccol = bspgemm.SpGEMM_hip(g["a_ci"], g["a_rp"], rows, g["a_ci"], g["a_rp"], n, tBlock=64, row0=r0)
    assert_same(crow, ccol, g["c_rp"], g["c_ci"])
    crow, ccol = bspgemm.SpGEMM_hip_bigslice(g["a_ci"], g["a_rp"], n, g["a_ci"], g["a_rp"], n, r0, r0 + rows)
    assert_same(crow, ccol, g["c_rp"], g["c_ci"])


def test_rect_golden_and_mat_dropin(ctx):
    g = np.load(os.path.join(GOLDEN, "rect_300x200x250_mat.npz"), allow_pickle=False)
    crp, cci, _ = hip_product(ctx, g["a_rp"], g["a_ci"], 200, g["b_rp"], g["b_ci"], 250)
    assert_same(crp, cci, g["c_rp"], g["c_ci"])
    crow, ccol = bspgemm.SpGEMM_hip_mat(g["a_ci"], g["a_rp"], 300, g["b_ci"], g["b_rp"], 250, g["c_rp"][-1])
    assert_same(crow, ccol, g["c_rp"], g["c_ci"])


# ---------------------------------------------------------------- (2) oracle, every path ---
CASES = {
    # name: (maker, expected properties)
    "levels1_n3000": lambda: gen.uniform(3000, 12, 201),                       # cols <= 8192
    "levels2_uniform_n2e16_d16": lambda: gen.uniform(1 << 16, 16, 202),        # cfg-2 shape, smaller
    "levels2_rmat_s15_mild": lambda: gen.rmat(15, 16, (0.30, 0.25, 0.25, 0.20), 203),
    "levels2_rmat_s14_g500": lambda: gen.rmat(14, 16, (0.57, 0.19, 0.19, 0.05), 204),   # hub rows -> dense kernel
    "levels3_uniform_n300k": lambda: gen.uniform(300000, 6, 205),              # cols > 2^18
    "levels3_rmat_s19_sparse": lambda: gen.rmat(19, 4, (0.45, 0.22, 0.22, 0.11), 206),
    "powerlaw_n2e15_d32": lambda: gen.powerlaw(1 << 15, 32, 207),              # cfg-5 shape, smaller
    "special_rows": lambda: gen.with_special_rows(5000, 7, 208),
    "dups_unsorted": lambda: gen.dups_unsorted(4097, 20, 209),
    "banded": lambda: gen.banded(70000, 9, 210),
    "n_not_multiple_of_64": lambda: gen.uniform(4095 + 64 * 3 + 1, 9, 211),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_against_oracle(ctx, name):
    rp, ci, n = CASES[name]()
    erp, eci = O.spgemm(rp, ci, rp, ci, n)
    crp, cci, st = hip_product(ctx, rp, ci, n, rp, ci, n)
    assert_same(crp, cci, erp, eci)
    assert st["products"] == O.count_products(rp, ci, rp)
    assert st["nnz_c"] == erp[-1] and sum(st["rows_per_bin"]) == n


def test_every_capacity_class_is_exercised(ctx):
    """rows with F_i just below/above each class boundary 64,128,...,2048, 524288 (the small heavy-row shape's cap
    when one window covers the columns) and beyond"""
    n = 6000
    rng = np.random.default_rng(301)
    # B: row j has (j % 97) + 1 entries; A rows pick B rows so that F_i sweeps 1..4000
    b_rows = np.repeat(np.arange(n), (np.arange(n) % 97) + 1)
    b_cols = rng.integers(0, n, size=b_rows.size)
    b_rp, b_ci = gen._csr_from_pairs(b_rows, b_cols, n)
    blen = np.diff(b_rp)
    caps = [64 * c for c in (1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32)]   # csrc/kernels.hpp kWaveChunks
    targets = ([1, 2] + [t for cap in caps for t in (cap - 1, cap, cap + 1)] + [3000, 4000, 100000, 280000, 600000]) * 4
    a_rows, a_cols = [], []
    for i, t in enumerate(targets):
        acc = 0
        while acc < t:
            j = int(rng.integers(0, n))
            if acc + blen[j] <= t + 3:
                a_rows.append(i)
                a_cols.append(j)
                acc += blen[j]
    # (repeated column entries stay: a 6000-column A row could not reach 600000 products without them --
    # rows need not be duplicate-free)
    a_rp, a_ci = gen._csr_from_pairs(a_rows, a_cols, n, dedup=False)
    erp, eci = O.spgemm(a_rp, a_ci, b_rp, b_ci, n)
    crp, cci, st = hip_product(ctx, a_rp, a_ci, n, b_rp, b_ci, n)
    assert_same(crp, cci, erp, eci)
    assert st["bin_cap"][1:-3] == caps and st["bins"] == len(caps) + 4, st["bin_cap"]
    RANK = len(caps) + 1                                      # (the rank class is for 2^18 < cols <= 2^20: test_rank_rows_class)
    assert all(c > 0 for k, c in enumerate(st["rows_per_bin"]) if k != RANK) and st["rows_per_bin"][RANK] == 0, st["rows_per_bin"]


@pytest.mark.parametrize("ncols", [12_000_000, 40_000_000, 300_000_000],
                         ids=["levels3_wide_top_12M", "levels4_40M", "levels5_300M"])
def test_wide_columns(ctx, ncols):
    """2^23 < cols <= 2^24 -> three levels under a 512-word top bitmap, cols > 2^24 -> four 5-bit
    levels, cols > 2^28 -> five; B is 2000 x ncols"""
    a_rp, a_ci = gen.uniform_rect(1500, 2000, 8, seed=401)
    rng = np.random.default_rng(402)
    rows = np.repeat(np.arange(2000), 30)
    cols = np.concatenate([rng.integers(0, ncols, size=rows.size // 2),
                           rng.integers(ncols - 5000, ncols, size=rows.size - rows.size // 2)])
    b_rp, b_ci = gen._csr_from_pairs(rows, cols, 2000)
    erp, eci = O.spgemm(a_rp, a_ci, b_rp, b_ci, ncols)
    crp, cci, _ = hip_product(ctx, a_rp, a_ci, 2000, b_rp, b_ci, ncols)
    assert_same(crp, cci, erp, eci)


def test_dense_rows_several_windows(ctx):
    """heavy rows with cols > 2^20: the dense kernel sweeps several LDS windows"""
    ncols = 3 * (1 << 20) + 777
    rng = np.random.default_rng(501)
    nb = 3000
    rows = np.repeat(np.arange(nb), 200)
    cols = rng.integers(0, ncols, size=rows.size)
    b_rp, b_ci = gen._csr_from_pairs(rows, cols, nb)
    a_rows = np.concatenate([np.zeros(2500, np.int64), np.full(40, 1), np.full(900, 2), np.arange(3, 60)])
    a_cols = rng.integers(0, nb, size=a_rows.size)
    a_rp, a_ci = gen._csr_from_pairs(a_rows, a_cols, 64)
    erp, eci = O.spgemm(a_rp, a_ci, b_rp, b_ci, ncols)
    crp, cci, st = hip_product(ctx, a_rp, a_ci, nb, b_rp, b_ci, ncols)
    assert_same(crp, cci, erp, eci)
    assert st["rows_per_bin"][-1] >= 2 and st["rows_per_bin"][-2] >= 1     # both heavy-row shapes, several windows each


@pytest.mark.parametrize("ncols", [300_000, 700_001, 1 << 20, 3 * (1 << 20) + 777, (1 << 24) - 5],
                         ids=["300k_one_top_word_per_thread", "700k", "2pow20", "3_spans_and_a_bit", "16_spans"])
def test_rank_rows_class(ctx, ncols):
    """rows of 2048 < F_i <= 6144 products over two to four windows of the small heavy-row shape take the rank class
    (csrc/dense_rows.hip k_rank_rows), on wider matrices in spans of 2^20 columns: boundaries of the class, dense column
    clusters (full 32-column slots, full top words), sources of one to three entries (masked quads), more than 512 sources
    (several batches) and more than 4096 quads (several tiles), repeated A entries, the last column"""
    rng = np.random.default_rng(ncols % 1000 + 77)
    nb = 9000
    lens = np.concatenate([rng.integers(1, 4, 6000), rng.integers(4, 200, 2000), rng.integers(200, 1500, 1000)])
    rows, cols = [], []
    for j, L in enumerate(lens):
        kind = j % 4
        if kind == 0:      # a dense cluster somewhere (consecutive columns: whole slots and top words)
            c0 = int(rng.integers(0, ncols - L))
            c = np.arange(c0, c0 + L)
        elif kind == 1:    # the tail of the column range, including the last column
            c = ncols - 1 - rng.choice(min(ncols, 4 * L + 8), size=L, replace=False)
        elif kind == 2 and ncols > (1 << 20):   # around a span boundary
            c = (1 << 20) * int(rng.integers(1, (ncols >> 20) + 1)) - 2 * L + rng.choice(4 * L, size=L, replace=False)
            c = c[c < ncols]
            c = np.concatenate([c, rng.choice(1000, size=L - c.size, replace=False)]) if c.size < L else c
        else:
            c = rng.permutation(np.unique(rng.integers(0, ncols, size=2 * L)))[:L] if ncols > (1 << 21) else rng.choice(ncols, size=L, replace=False)
            c = np.concatenate([c, ncols - 1 - np.arange(L - c.size)]) if c.size < L else rng.permutation(c)
        rows.append(np.full(L, j)); cols.append(c)
    b_rp, b_ci = gen._csr_from_pairs(np.concatenate(rows), np.concatenate(cols), nb)
    blen = np.diff(b_rp)
    short = np.flatnonzero(blen <= 3); longer = np.flatnonzero(blen > 3)
    a_rows, a_cols = [], []
    targets = [2049, 2100, 3000, 4097, 5000, 6143, 6144, 6145, 7000, 2500, 2049, 6144, 5000]
    ones = np.flatnonzero(blen == 1)
    for i, t in enumerate(targets):
        acc = 0
        pool = short if i in (3, 4, 9) else longer            # rows 3, 4, 9: thousands of one-to-three-entry sources
        if i == 12:
            pool = ones                                       # 5000 one-entry sources: 5000 quads, two tiles of the gather plan
        while acc < t:
            j = int(pool[rng.integers(0, pool.size)])
            if acc + blen[j] <= t:
                a_rows.append(i); a_cols.append(j); acc += blen[j]
            elif t - acc <= 3:
                j = int(short[np.flatnonzero(blen[short] == t - acc)[0]])
                a_rows.append(i); a_cols.append(j); acc += blen[j]
    a_rp, a_ci = gen._csr_from_pairs(a_rows, a_cols, len(targets), dedup=False)
    erp, eci = O.spgemm(a_rp, a_ci, b_rp, b_ci, ncols)
    crp, cci, st = hip_product(ctx, a_rp, a_ci, nb, b_rp, b_ci, ncols)
    assert_same(crp, cci, erp, eci)
    F = np.array([blen[a_ci[a_rp[i]:a_rp[i + 1]]].sum() for i in range(len(targets))])
    assert list(F) == targets
    RANK = st["bins"] - 3
    assert st["bin_cap"][RANK] == 6144 and st["rows_per_bin"][RANK] == sum(1 for t in targets if t <= 6144), st["rows_per_bin"]
    assert st["rows_per_bin"][RANK + 1] == sum(1 for t in targets if t > 6144)


@pytest.mark.parametrize("nnzb", [1, 2, 3, 5])
def test_heavy_row_over_a_tiny_b(ctx, nnzb):
    """the heavy rows gather B in 16-byte quads; a B.col_idx of fewer than four entries has no room for one (scalar loads),
    and the quads of the first sources of a B begin before the array (clamped and masked): a row of 3000-5000 repeated A entries
    over a B of 1, 2, 3 and 5 nonzeros"""
    ncols = 11
    b_rows = [0, 1, 1, 2, 2][:nnzb]
    b_cols = [7, 0, 10, 3, 4][:nnzb]
    b_rp, b_ci = gen._csr_from_pairs(b_rows, b_cols, 3)
    rng = np.random.default_rng(900 + nnzb)
    nrep = 5000
    a_rows = np.concatenate([np.zeros(nrep, np.int64), np.ones(3, np.int64)])
    a_cols = np.concatenate([rng.integers(0, 3, size=nrep), [0, 1, 2]])
    a_rp, a_ci = gen._csr_from_pairs(a_rows, a_cols, 2, dedup=False)
    erp, eci = O.spgemm(a_rp, a_ci, b_rp, b_ci, ncols)
    crp, cci, st = hip_product(ctx, a_rp, a_ci, 3, b_rp, b_ci, ncols)
    assert_same(crp, cci, erp, eci)
    if nnzb >= 2:
        assert sum(st["rows_per_bin"][-3:]) == 1, st["rows_per_bin"]      # row 0 is a heavy row (F > 2048)


def test_fuzz_small_shapes(ctx):
    """150 seeded random shapes against the oracle: rectangular, skewed, duplicate-heavy, column
    counts on every side of the level boundaries (8192, 2^18, 2^23, 2^24), row ranges, masks"""
    rng = np.random.default_rng(20261004)
    col_choices = [1, 7, 64, 65, 1000, 8192, 8193, 50_000, 262_144, 262_145, 3_000_000, 8_388_608, 8_388_609,
                   16_777_216, 16_777_217, 100_000_000]
    for case in range(150):
        ar = int(rng.integers(1, 400))
        inner = int(rng.integers(1, 400))
        ncols = int(col_choices[int(rng.integers(0, len(col_choices)))])
        da = int(rng.integers(0, 12))
        db = int(rng.choice([0, 1, 3, 17, 80, 300]))
        # A: ar x inner, B: inner x ncols; a few hub rows in both
        a_rows = np.concatenate([np.repeat(np.arange(ar), da), np.zeros(int(rng.integers(0, 200)), np.int64)])
        a_cols = rng.integers(0, inner, size=a_rows.size)
        b_rows = np.concatenate([np.repeat(np.arange(inner), db), np.full(int(rng.integers(0, 3000)), inner - 1)])
        span = ncols if rng.random() < 0.5 else min(ncols, 97)          # narrow span -> many duplicates
        b_cols = rng.integers(0, span, size=b_rows.size)
        a_rp, a_ci = gen._csr_from_pairs(a_rows, a_cols, ar, dedup=bool(case & 1))
        b_rp, b_ci = gen._csr_from_pairs(b_rows, b_cols, inner, dedup=bool(case & 2))
        r0 = int(rng.integers(0, ar)) if case % 5 == 0 else 0
        r1 = int(rng.integers(r0, ar + 1)) if case % 5 == 0 else ar
        erp, eci = O.spgemm_rows(a_rp, a_ci, b_rp, b_ci, ncols, r0, r1)
        crp, cci, _ = hip_product(ctx, a_rp, a_ci, inner, b_rp, b_ci, ncols, r0, r1)
        assert_same(crp, cci, erp, eci)
        if case % 4 == 0:                                               # masked with a random mask
            f_rows = np.repeat(np.arange(ar), int(rng.integers(1, 40)))
            f_cols = rng.integers(0, span, size=f_rows.size)
            f_rp, f_ci = gen._csr_from_pairs(f_rows, f_cols, ar)
            mrp, mci = O.spgemm_masked(a_rp, a_ci, b_rp, b_ci, ncols, f_rp, f_ci)
            A = ctx.upload(a_rp, a_ci, inner)
            B = ctx.upload(b_rp, b_ci, ncols)
            Fm = ctx.upload(f_rp, f_ci, ncols)
            C = ctx.multiply_masked(A, B, Fm)
            grp, gci = C.download()
            assert_same(grp, gci, mrp, mci)
            for h in (C, A, B, Fm):
                h.free()


@pytest.mark.parametrize("streams", [1, 2, 3], ids=["one_stream", "two_streams", "three_streams"])
def test_tuning_knobs_do_not_change_results(ctx, streams):
    """the stream knob only reorders launches.  Set through the ABI (bspgemm_set_option) -- the
    environment variable of the same name is read once, in bspgemm_create -- and the stats of the multiply must say
    that many streams were used, so the test cannot go vacuous again."""
    rp, ci, n = gen.uniform(300_000, 4, 915)
    erp, eci = O.spgemm(rp, ci, rp, ci, n)
    old = ctx.get_option("class_streams")
    ctx.set_option("class_streams", streams)
    try:
        assert ctx.get_option("class_streams") == streams
        crp, cci, st = hip_product(ctx, rp, ci, n, rp, ci, n)
    finally:
        ctx.set_option("class_streams", old)
    assert st["class_streams"] == streams
    assert_same(crp, cci, erp, eci)
    with pytest.raises(bspgemm.BspgemmError):
        ctx.set_option("class_streams", 4)


def test_env_knobs_are_read_at_create():
    """the environment variables still work -- for a context created AFTER they are set"""
    old = {k: os.environ.get(k) for k in ("BSPGEMM_CLASS_STREAMS", "BSPGEMM_RW_BLK", "BSPGEMM_FLOW")}
    os.environ.update({"BSPGEMM_CLASS_STREAMS": "3", "BSPGEMM_RW_BLK": "1", "BSPGEMM_FLOW": "exact"})
    try:
        c = bspgemm.Context(0)
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
    try:
        assert c.get_option("class_streams") == 3 and c.get_option("blocked_extents") == 1
        rp, ci, n = gen.uniform(2000, 5, 916)
        erp, eci = O.spgemm(rp, ci, rp, ci, n)
        A = c.upload(rp, ci, n)
        C = c.multiply(A, A)
        crp, cci = C.download()
        st = c.stats()
        assert st["flow"] == 2 and st["class_streams"] == 3 and st["prepass_kernel"] == 1 and A.uses_blocked_table == 1
        assert_same(crp, cci, erp, eci)
    finally:
        c.close()


def test_small_product_path(ctx):
    """csrc/small.hip: a product with few A-nonzeros takes five launches and ONE host round trip.
    bspgemm_stats.small_path says whether it ran; switched off (BSPGEMM_OPT_SMALL_PATH = 0) the general flow must give the
    same CSR; a product that looks small to the host but does not fit (a row of more than 2048 products, or more than
    65536 products in all) is found out on the device and re-run through the general flow -- same result again."""
    exact = ctx.flow_name == "exact"                       # (the exact flow is never replaced: it is the flow under test there)
    rp, ci, m, n = bspgemm.readCOO(os.path.join(GOLDEN, "validity_test.mtx"))
    g = np.load(os.path.join(GOLDEN, "validity.npz"), allow_pickle=False)
    old = ctx.get_option("small_path")
    try:
        for mode, want in ((-1, 0 if exact else 1), (0, 0), (1, 0 if exact else 1)):
            ctx.set_option("small_path", mode)
            crp, cci, st = hip_product(ctx, rp, ci, n, rp, ci, n)
            assert st["small_path"] == want, (mode, st["small_path"])
            assert st["products"] == 12502 and st["nnz_c"] == 12502 and sum(st["rows_per_bin"]) == n
            assert_same(crp, cci, g["c_rp"], g["c_ci"])
        # unsorted rows with duplicates, rectangular, an interior row range: the sort + squeeze of one wave per row
        ctx.set_option("small_path", 1)
        a_rp, a_ci, _ = gen.dups_unsorted(900, 6, 941)
        rng = np.random.default_rng(942)
        b_rows = np.repeat(np.arange(900 // 8 + 1), 9)
        b_rp, b_ci = gen._csr_from_pairs(b_rows, rng.integers(0, 5000, size=b_rows.size), 900 // 8 + 1, dedup=False, sort=False)
        erp, eci = O.spgemm_rows(a_rp, a_ci, b_rp, b_ci, 5000, 100, 777)
        crp, cci, st = hip_product(ctx, a_rp, a_ci, 900 // 8 + 1, b_rp, b_ci, 5000, 100, 777)
        assert st["small_path"] == (0 if exact else 1)
        assert_same(crp, cci, erp, eci)
        # looks small (40 A-nonzeros) but one row has 40 * 200 = 8000 products: the device says "does not fit"
        a_rp2 = np.array([0, 40, 41, 41], np.int32)
        a_ci2 = np.concatenate([np.arange(40), [3]]).astype(np.int32)
        b_rows = np.repeat(np.arange(50), 200)
        b_rp2, b_ci2 = gen._csr_from_pairs(b_rows, rng.integers(0, 100_000, size=b_rows.size), 50)
        erp, eci = O.spgemm(a_rp2, a_ci2, b_rp2, b_ci2, 100_000)
        crp, cci, st = hip_product(ctx, a_rp2, a_ci2, 50, b_rp2, b_ci2, 100_000)
        assert st["small_path"] == 0 and st["products"] > 2048
        assert_same(crp, cci, erp, eci)
        # ... and more than 65536 products in all from 400 A-nonzeros
        a_rp3, a_ci3 = gen.uniform_rect(200, 50, 2, seed=943)
        erp, eci = O.spgemm(a_rp3, a_ci3, b_rp2, b_ci2, 100_000)
        crp, cci, st = hip_product(ctx, a_rp3, a_ci3, 50, b_rp2, b_ci2, 100_000)
        assert st["small_path"] == 0 and st["products"] > 65536
        assert_same(crp, cci, erp, eci)
    finally:
        ctx.set_option("small_path", old)


def test_class_timing_switch(ctx):
    """per-class event brackets are off by default (they cost ~1 % of a large product), on request the stats carry them;
    the result is the same either way and the phase times are always there"""
    rp, ci, n = gen.rmat(13, 16, (0.57, 0.19, 0.19, 0.05), 11)
    A = ctx.upload(rp, ci, n)
    C = ctx.multiply(A, A); ref = C.download(); C.free()
    st = ctx.stats()
    assert float(np.sum(st["ms_bin"])) == 0.0 and st["ms_numeric"] > 0 and st["ms_total"] > 0
    ctx.set_class_timing(True)
    try:
        C = ctx.multiply(A, A); got = C.download(); C.free()
        st = ctx.stats()
        assert float(np.sum(st["ms_bin"])) + float(np.sum(st["ms_bin_count"])) > 0.0
    finally:
        ctx.set_class_timing(False)
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])


def test_blocked_extents_table_forced(ctx):
    """k_row_work_blk (csrc/prepass.hip) is chosen per operand only for B of 2^21 rows or more; forced here
    (bspgemm_set_option BLOCKED_EXTENTS = 1, checked through bspgemm_stats.prepass_kernel and
    bspgemm_matrix_uses_blocked_table) on small shapes that hit every branch of it: rows of 255+ nonzeros
    (clamped bytes -> exact lookup), a block whose clamped byte lies below / at / above the looked-up row, the
    last partial block, rectangular A != B, an interior row range, the masked product (the same prepass).
    The same inputs with the table forbidden (0) must give the same result through k_row_work."""
    rng = np.random.default_rng(4242)
    # B: 1003 rows (last block partial); hubs of 255, 256, 300 and 5000 nonzeros at block offsets 0, 3 and 7
    lens = rng.integers(0, 40, size=1003)
    for r, L in ((0, 255), (11, 256), (23, 300), (512, 5000), (1002, 700), (1000, 254)):
        lens[r] = L
    ncols = 9000
    b_rows = np.repeat(np.arange(1003), lens)
    b_cols = rng.integers(0, ncols, size=b_rows.size)
    b_rp, b_ci = gen._csr_from_pairs(b_rows, b_cols, 1003, dedup=False)       # exact lengths 254/255/256
    a_rows = np.repeat(np.arange(700), rng.integers(0, 30, size=700))
    a_cols = rng.integers(0, 1003, size=a_rows.size)
    a_cols[:64] = np.array([0, 1, 7, 8, 11, 12, 15, 16, 23, 24, 512, 513, 519, 1000, 1001, 1002] * 4)
    a_rp, a_ci = gen._csr_from_pairs(a_rows, a_cols, 700)
    erp, eci = O.spgemm(a_rp, a_ci, b_rp, b_ci, ncols)
    rp, ci, n = bspgemm.gen_powerlaw(20_000, 24, seed=77)
    rp, ci = np.asarray(rp), np.asarray(ci)
    assert np.diff(rp).max() >= 255
    prp, pci = O.spgemm(rp, ci, rp, ci, n)
    frp, fci = O.spgemm_masked(rp, ci, rp, ci, n, rp, ci)
    old = ctx.get_option("blocked_extents")
    try:
        for force in (1, 0):
            ctx.set_option("blocked_extents", force)
            A = ctx.upload(a_rp, a_ci, 1003)
            B = ctx.upload(b_rp, b_ci, ncols)
            assert B.uses_blocked_table == -1                     # not decided before its first use as B
            C = ctx.multiply(A, B)
            crp, cci = C.download()
            st = ctx.stats()
            assert st["prepass_kernel"] == force and B.uses_blocked_table == force, (force, st["prepass_kernel"])
            assert_same(crp, cci, erp, eci)
            for h in (C, A, B):
                h.free()
            # interior rows of a square power-law product + its masked form
            A = ctx.upload(rp, ci, n)
            C = ctx.multiply(A, A, 3000, 17000)
            assert ctx.stats()["prepass_kernel"] == force
            grp, gci = C.download()
            assert np.array_equal(grp, prp[3000:17001] - prp[3000]) and np.array_equal(gci, pci[prp[3000]:prp[17000]])
            M = ctx.multiply_masked(A, A, A)
            assert ctx.stats()["prepass_kernel"] == force
            mrp, mci = M.download()
            assert_same(mrp, mci, frp, fci)
            for h in (C, M, A):
                h.free()
    finally:
        ctx.set_option("blocked_extents", old)


def test_blocked_extents_table_at_its_real_size(ctx):
    """The prepass of the benchmarked flow where it is chosen BY DEFAULT (B of 2^21 rows or more): a mid-skew R-MAT
    (0.45, 0.22, 0.22, 0.11) of scale 21 has B rows of 255+ nonzeros (clamped bytes -> the SWAR saturation test and the
    exact B.row_ptr look-up of k_row_work_blk) and, with n - 3 rows, a partial last block of the table (k_blk8's
    tail).  Compared completely against the oracle (this branch ran in no test of GPUTEST_r03)."""
    rp, ci, n = bspgemm.gen_rmat(21, 4, (0.45, 0.22, 0.22), seed=5)
    rp, ci = np.asarray(rp), np.asarray(ci)
    # five more rows and columns (n + 5 is not a multiple of 8: the table's last block is partial); the new rows point
    # at the hubs (rows of 255+ entries) and at each other
    rows = n + 5
    hubs = np.argsort(np.diff(rp))[-3:].astype(np.int32)
    extra = [np.sort(np.concatenate([hubs, [n + (k + 1) % 5, n + (k + 3) % 5]])).astype(np.int32) for k in range(5)]
    sub_ci = np.concatenate([ci] + extra)
    sub_rp = np.concatenate([rp, rp[-1] + 5 * np.arange(1, 6, dtype=np.int32)]).astype(np.int32)
    deg = np.diff(sub_rp)
    assert rows >= (1 << 21) and rows % 8 != 0 and deg.max() >= 255, (rows, deg.max())
    assert ctx.get_option("blocked_extents") == -1     # the per-operand decision, not a forced one
    A = ctx.upload(sub_rp, sub_ci, rows)
    C = ctx.multiply(A, A)
    st = ctx.stats()
    assert st["prepass_kernel"] == 1 and A.uses_blocked_table == 1, "the blocked table was not chosen for a 2^21-row B"
    crp, cci = C.download()
    C.free()
    A.free()
    erp, eci = O.spgemm_omp(sub_rp, sub_ci, sub_rp, sub_ci, rows, 4096, 0)
    assert_same(crp, cci, erp, eci)
    assert st["products"] == O.count_products(sub_rp, sub_ci, sub_rp)
    # and A-nonzeros that point at clamped rows exist (the branch is really taken)
    assert int((deg[sub_ci] >= 255).sum()) > 1000


def test_padded_rows_forced(ctx):
    """BSPGEMM_OPT_PADDED_ROWS: the accumulate kernels gather B's rows from a derived copy of B.col_idx in which every row
    starts on a 64-byte boundary (csrc/context.hip ensure_pad, csrc/prepass.hip k_pad_copy); chosen per operand only for
    large B, forced here -- with and without the blocked extents table (the two prepass kernels that produce padded
    starts) -- on shapes with empty rows, rows of 1, 15, 16, 17 and 255+ entries, unsorted rows with duplicates,
    A != B, an interior row range, heavy rows and the masked product.  bspgemm_stats.padded_rows says that it ran."""
    rng = np.random.default_rng(961)
    lens = rng.integers(0, 40, size=1500)
    for r, L in ((0, 16), (1, 15), (2, 17), (3, 1), (4, 0), (5, 255), (6, 256), (7, 700), (8, 32), (9, 33), (1499, 48)):
        lens[r] = L
    ncols = 20000
    b_rows = np.repeat(np.arange(1500), lens)
    b_rp, b_ci = gen._csr_from_pairs(b_rows, rng.integers(0, ncols, size=b_rows.size), 1500, dedup=False, sort=False)
    a_rows = np.concatenate([np.repeat(np.arange(900), rng.integers(0, 25, size=900)), np.full(400, 77), np.full(1400, 78)])
    a_cols = rng.integers(0, 1500, size=a_rows.size)
    a_cols[:16] = np.arange(16) % 10
    a_rp, a_ci = gen._csr_from_pairs(a_rows, a_cols, 900)
    erp, eci = O.spgemm(a_rp, a_ci, b_rp, b_ci, ncols)
    srp, sci, sn = gen.rmat(14, 16, (0.57, 0.19, 0.19, 0.05), 962)        # square, skewed: every class incl. heavy rows
    prp, pci = O.spgemm(srp, sci, srp, sci, sn)
    mrp0, mci0 = O.spgemm_masked(srp, sci, srp, sci, sn, srp, sci)
    old_pad, old_blk = ctx.get_option("padded_rows"), ctx.get_option("blocked_extents")
    try:
        for blk in (0, 1):
            ctx.set_option("blocked_extents", blk)
            ctx.set_option("padded_rows", 1)
            A = ctx.upload(a_rp, a_ci, 1500)
            B = ctx.upload(b_rp, b_ci, ncols)
            assert B.uses_padded_rows == -1
            C = ctx.multiply(A, B)
            st = ctx.stats()
            assert st["padded_rows"] == 1 and B.uses_padded_rows == 1 and st["prepass_kernel"] == blk
            crp, cci = C.download()
            assert_same(crp, cci, erp, eci)
            for h in (C, A, B):
                h.free()
            S = ctx.upload(srp, sci, sn)
            C = ctx.multiply(S, S, 1000, 15000)
            assert ctx.stats()["padded_rows"] == 1
            grp, gci = C.download()
            assert np.array_equal(grp, prp[1000:15001] - prp[1000]) and np.array_equal(gci, pci[prp[1000]:prp[15000]])
            M = ctx.multiply_masked(S, S, S)
            assert ctx.stats()["padded_rows"] == 1
            mrp, mci = M.download()
            assert_same(mrp, mci, mrp0, mci0)
            for h in (C, M, S):
                h.free()
        # switched off: the same operands through B.col_idx itself
        ctx.set_option("padded_rows", 0)
        A = ctx.upload(a_rp, a_ci, 1500)
        B = ctx.upload(b_rp, b_ci, ncols)
        C = ctx.multiply(A, B)
        assert ctx.stats()["padded_rows"] == 0 and B.uses_padded_rows == 0
        crp, cci = C.download()
        assert_same(crp, cci, erp, eci)
        for h in (C, A, B):
            h.free()
    finally:
        ctx.set_option("padded_rows", old_pad)
        ctx.set_option("blocked_extents", old_blk)


def test_rewritten_operand_needs_invalidate(ctx):
    """bspgemm_matrix_wrap_device: the library keeps tables derived from row_ptr (byte lengths, blocked extents).  A
    caller that rewrites the wrapped arrays in place and calls bspgemm_matrix_invalidate gets the new product; one
    that does not is caught by the debug check (BSPGEMM_OPT_CHECK): BSPGEMM_ERR_INVALID instead of rows sized from
    stale lengths."""
    import torch
    dev = torch.device("cuda", 0)
    n = 6000
    rp1, ci1, _ = gen.uniform(n, 6, 931)
    rp2, ci2, _ = gen.powerlaw(n, 6, 932)                 # same n, other row lengths (some of 255+)
    cap = max(ci1.size, ci2.size)
    d_rp = torch.from_numpy(np.asarray(rp1, np.int32)).to(dev)
    d_ci = torch.zeros(cap, dtype=torch.int32, device=dev)
    d_ci[: ci1.size] = torch.from_numpy(np.asarray(ci1, np.int32)).to(dev)
    old_blk, old_chk, old_pad = ctx.get_option("blocked_extents"), ctx.get_option("check"), ctx.get_option("padded_rows")
    ctx.set_option("blocked_extents", 1)                  # every derived table in play: byte lengths, blocked extents,
    ctx.set_option("padded_rows", 1)                      # the padded copy of col_idx
    try:
        A = ctx.wrap_device(n, n, int(ci1.size), d_rp.data_ptr(), d_ci.data_ptr(), keep=(d_rp, d_ci))
        C = ctx.multiply(A, A)
        crp, cci = C.download(); C.free()
        erp, eci = O.spgemm(rp1, ci1, rp1, ci1, n)
        assert_same(crp, cci, erp, eci)
        assert A.uses_blocked_table == 1
        # rewrite in place
        d_rp.copy_(torch.from_numpy(np.asarray(rp2, np.int32)))
        d_ci[: ci2.size] = torch.from_numpy(np.asarray(ci2, np.int32)).to(dev)
        torch.cuda.synchronize()
        # (1) without invalidate the debug check refuses
        ctx.set_option("check", 1)
        with pytest.raises(bspgemm.BspgemmError) as e:
            ctx.multiply(A, A)
        assert e.value.status == 1 and "invalidate" in str(e.value)          # BSPGEMM_ERR_INVALID
        # (2) with invalidate: the new product, tables rebuilt, the check is satisfied
        A.invalidate()
        assert A.uses_blocked_table == -1 and A.uses_padded_rows == -1
        C = ctx.multiply(A, A)
        st = ctx.stats()
        assert st["checked"] == 1 and st["prepass_kernel"] == 1 and st["padded_rows"] == 1
        crp, cci = C.download(); C.free()
        erp, eci = O.spgemm(rp2, ci2, rp2, ci2, n)
        assert_same(crp, cci, erp, eci)
        A.free()
    finally:
        ctx.set_option("check", old_chk)
        ctx.set_option("blocked_extents", old_blk)
        ctx.set_option("padded_rows", old_pad)


def test_mostly_empty_rows(ctx):
    """a result whose 32768-output compaction chunks span far more than 4096 rows (every 40th row
    is non-empty): the per-output row search of k_compact; also the masked product of the same shape"""
    n = 400_000
    rng = np.random.default_rng(907)
    live = np.arange(0, n, 40)
    rows = np.repeat(live, 3)
    cols = rng.choice(live, size=rows.size)                 # products land on non-empty rows of B = A
    rp, ci = gen._csr_from_pairs(rows, cols, n)
    erp, eci = O.spgemm(rp, ci, rp, ci, n)
    crp, cci, st = hip_product(ctx, rp, ci, n, rp, ci, n)
    assert_same(crp, cci, erp, eci)
    assert 0 < erp[-1] < 200_000 and st["rows_per_bin"][0] > 0.97 * n
    A = ctx.upload(rp, ci, n)
    C = ctx.multiply_masked(A, A, A)
    mrp, mci = C.download()
    frp, fci = O.spgemm_masked(rp, ci, rp, ci, n, rp, ci)
    assert_same(mrp, mci, frp, fci)
    C.free()
    A.free()


def test_empty_and_degenerate(ctx):
    z = np.zeros(65, np.int32)
    crp, cci, _ = hip_product(ctx, z, np.zeros(0, np.int32), 64, z, np.zeros(0, np.int32), 64)
    assert crp.tolist() == [0] * 65 and cci.size == 0
    one_rp, one_ci = np.array([0, 1], np.int32), np.array([0], np.int32)
    crp, cci, _ = hip_product(ctx, one_rp, one_ci, 1, one_rp, one_ci, 1)
    assert crp.tolist() == [0, 1] and cci.tolist() == [0]
    rp, ci, n = gen.uniform(500, 5, 601)
    A = ctx.upload(rp, ci, n)
    C = ctx.multiply(A, A, 100, 100)          # empty row range
    assert C.rows == 0 and C.nnz == 0
    with pytest.raises(bspgemm.BspgemmError):
        ctx.multiply(A, A, 10, 5)
    with pytest.raises(bspgemm.BspgemmError):
        ctx.multiply(A, A, 0, n + 1)


# ---------------------------------------------------------------- masked product ----------
def test_masked_golden_and_dropin(mctx):
    ctx = mctx
    """C = F .* (A*B) against the reference's SpGEMM_masked golden (final/SpGEMM_mpi_omp.c:232-288)"""
    g = np.load(os.path.join(GOLDEN, "masked_n512.npz"), allow_pickle=False)
    n = int(g["n"])
    A = ctx.upload(g["a_rp"], g["a_ci"], n)
    F = ctx.upload(g["f_rp"], g["f_ci"], n)
    C = ctx.multiply_masked(A, A, F)
    crp, cci = C.download()
    assert_same(crp, cci, g["c_rp"], g["c_ci"])
    crow, ccol = bspgemm.SpGEMM_hip_masked(g["a_ci"], g["a_rp"], n, g["a_ci"], g["a_rp"], n, g["f_ci"], g["f_rp"])
    assert_same(crow, ccol, g["c_rp"], g["c_ci"])


@pytest.mark.parametrize("ncols", [5000, 100_000, 700_000], ids=["levels1", "levels2", "levels3_two_windows"])
def test_masked_against_oracle(mctx, ncols):
    ctx = mctx
    """rectangular A != B, mask rows of mixed length (empty, short, dense), row sub-range"""
    a_rp, a_ci = gen.uniform_rect(900, 700, 9, seed=801)
    rng = np.random.default_rng(802)
    rows = np.repeat(np.arange(700), 40)
    b_rp, b_ci = gen._csr_from_pairs(rows, rng.integers(0, ncols, size=rows.size), 700)
    frows = np.concatenate([np.repeat(np.arange(0, 900, 3), 300), np.repeat(np.arange(1, 900, 3), 5), np.full(ncols // 2, 7)])
    fcols = np.concatenate([rng.integers(0, ncols, size=frows.size - ncols // 2), np.arange(0, ncols // 2 * 2, 2)])
    f_rp, f_ci = gen._csr_from_pairs(frows, fcols, 900)
    erp, eci = O.spgemm_masked(a_rp, a_ci, b_rp, b_ci, ncols, f_rp, f_ci)
    A = ctx.upload(a_rp, a_ci, 700)
    B = ctx.upload(b_rp, b_ci, ncols)
    F = ctx.upload(f_rp, f_ci, ncols)
    C = ctx.multiply_masked(A, B, F)
    crp, cci = C.download()
    assert_same(crp, cci, erp, eci)
    C2 = ctx.multiply_masked(A, B, F, 100, 433)                 # interior rows keep F aligned by row id
    r2, c2 = C2.download()
    assert np.array_equal(r2, erp[100:434] - erp[100]) and np.array_equal(c2, eci[erp[100]:erp[433]])


def test_masked_triangle_pattern(mctx):
    ctx = mctx
    """C = A .* (A*A) on a skewed graph (the triangle-counting use of the masked product): mask rows
    of every capacity class, product counts far above the mask lengths (products are streamed)"""
    rp, ci, n = gen.rmat(14, 16, (0.57, 0.19, 0.19, 0.05), 811)
    erp, eci = O.spgemm_masked(rp, ci, rp, ci, n, rp, ci)
    A = ctx.upload(rp, ci, n)
    C = ctx.multiply_masked(A, A, A)
    crp, cci = C.download()
    assert_same(crp, cci, erp, eci)
    st = ctx.stats()
    assert st["products"] == O.count_products(rp, ci, rp) and st["nnz_c"] == erp[-1]


# ---------------------------------------------------------------- (3) properties at size --
def _check_wellformed(rp, ci, ncols):
    assert rp[0] == 0 and rp[-1] == ci.size and np.all(np.diff(rp) >= 0)
    if ci.size:
        assert ci.min() >= 0 and ci.max() < ncols
        d = np.diff(ci.astype(np.int64))
        starts = rp[1:-1][(rp[1:-1] > 0) & (rp[1:-1] < ci.size)]
        boundary = np.zeros(ci.size - 1, dtype=bool)
        boundary[starts - 1] = True
        assert np.all(d[~boundary] > 0), "col_idx must be strictly ascending inside every row"


def test_baseline_cfg2_uniform_full_size(ctx):
    """BASELINE config 2 at full size: n=2^18, d=16, A*A; exact vs the oracle (a few seconds)."""
    rp, ci, n = bspgemm.gen_uniform(1 << 18, 16, seed=1)
    crp, cci, st = hip_product(ctx, rp, ci, n, rp, ci, n)
    _check_wellformed(crp, cci, n)
    erp, eci = O.spgemm_omp(rp, ci, rp, ci, n, 4096, 0)
    assert_same(crp, cci, erp, eci)
    assert st["products"] == O.count_products(rp, ci, rp)


def test_shards_concatenate_to_whole(ctx):
    """row shards cut at equal work, multiplied separately, stitched == the whole product"""
    rp, ci, n = bspgemm.gen_rmat(17, 16, (0.30, 0.25, 0.25), seed=3)
    A = ctx.upload(rp, ci, n)
    whole = ctx.multiply(A, A)
    wrp, wci = whole.download()
    bounds = ctx.partition_rows(A, A, 4)
    assert bounds[0] == 0 and bounds[-1] == n and np.all(np.diff(bounds) >= 0)
    prefix = ctx.row_work_prefix(A, A)
    from bspgemm import dist as bdist
    assert np.array_equal(bounds, bdist.shard_bounds(prefix, 4))
    work = np.diff(prefix[bounds])
    assert work.max() < 1.2 * work.mean(), "equal-work cut is unbalanced: %s" % work
    parts_rp, parts_ci, base = [np.zeros(1, np.int64)], [], 0
    for p in range(4):
        C = ctx.multiply(A, A, int(bounds[p]), int(bounds[p + 1]))
        r, c = C.download()
        parts_rp.append(r[1:] + base)
        parts_ci.append(c)
        base += r[-1]
    assert_same(np.concatenate(parts_rp), np.concatenate(parts_ci), wrp, wci)
    _check_wellformed(wrp, wci, n)


def test_rmat_scale20_properties_and_sampled_rows(ctx):
    """R-MAT scale 20 (cfg-3 shape, 1/4 size): well-formedness + exact check of sampled row blocks"""
    rp, ci, n = bspgemm.gen_rmat(20, 16, (0.30, 0.25, 0.25), seed=1)
    A = ctx.upload(rp, ci, n)
    C = ctx.multiply(A, A)
    crp, cci = C.download()
    st = ctx.stats()
    _check_wellformed(crp, cci, n)
    assert st["products"] == O.count_products(rp, ci, rp)
    for r0 in (0, 1000, n // 2, n - 4096):
        erp, eci = O.spgemm_rows(rp, ci, rp, ci, n, r0, r0 + 4096)
        assert np.array_equal(crp[r0:r0 + 4097] - crp[r0], erp)
        assert np.array_equal(cci[crp[r0]:crp[r0 + 4096]], eci)


def test_baseline_cfg3_rmat_scale22_full_size(ctx):
    """BASELINE config 3 -- the matrix bench.py times (R-MAT scale 22, edge factor 16, mild skew, A*A,
    nnz(C) = 1.336 G) -- at FULL size, checked completely: well-formedness on the device, product
    count against the oracle, then row_ptr and all 1.3 G column indices against the oracle's
    OpenMP restatement of SpGEMM_omp (final/SpGEMM_mpi_omp.c:71-143) run on the host cores."""
    import torch
    from bspgemm import dist as bdist
    dev = torch.device("cuda", 0)
    rp, ci, n = bspgemm.gen_rmat(22, 16, (0.30, 0.25, 0.25), seed=1)
    A = ctx.upload(rp, ci, n)
    C = ctx.multiply(A, A)
    st = ctx.stats()
    crp, _ = C.download(col_idx=False)
    assert st["products"] == O.count_products(rp, ci, rp) == 1336443900
    assert st["padded_rows"] == 0 and st["prepass_kernel"] == 1       # what the bench line runs: the blocked table over B.col_idx itself
    assert C.nnz == crp[-1] == st["nnz_c"] == 1336366087          # the number bench.py prints
    assert crp[0] == 0 and np.all(np.diff(crp) >= 0)
    cols = bdist.device_tensor(C.col_idx_device, C.nnz, torch.int32, dev)
    assert int(cols.min()) >= 0 and int(cols.max()) < n
    bad = (cols[1:] <= cols[:-1])                                   # not ascending ... unless a new row starts there
    starts = torch.from_numpy(crp[1:-1][(crp[1:-1] > 0) & (crp[1:-1] < C.nnz)]).to(dev)
    bad[starts - 1] = False
    assert not bool(bad.any()), "col_idx must be strictly ascending inside every row"
    del bad, starts
    # blocks of 4096 rows, exact (cheap, and they localise a failure before the full comparison)
    for r0 in (0, 1000, n // 2, n - 4096):
        erp, eci = O.spgemm_rows(rp, ci, rp, ci, n, r0, r0 + 4096)
        assert np.array_equal(crp[r0:r0 + 4097] - crp[r0], erp)
        assert np.array_equal(cols[int(crp[r0]): int(crp[r0 + 4096])].cpu().numpy(), eci)
    del cols
    # everything
    _, cci = C.download()
    C.free()
    erp, eci = O.spgemm_omp(rp, ci, rp, ci, n, 4096, 0)
    assert np.array_equal(crp, erp)
    assert np.array_equal(cci, eci)


def _sampled_rows_exact(rp, ci, n, crp, col_tensor, starts, block=64):
    for r0 in starts:
        erp, eci = O.spgemm_rows(rp, ci, rp, ci, n, r0, r0 + block)
        assert np.array_equal(crp[r0:r0 + block + 1] - crp[r0], erp), r0
        got = col_tensor[int(crp[r0]): int(crp[r0 + block])].cpu().numpy()
        assert np.array_equal(got, eci), r0


def test_row_ranges_with_rank_and_windowed_heavy_rows(ctx):
    """interior row ranges of a skewed product over 2^19 columns (rank class, two windows of the small shape, the large
    shape): every range compared completely with the oracle, and the ranges add up to the whole"""
    rp, ci, n = bspgemm.gen_rmat(19, 3, (0.57, 0.19, 0.19), seed=11)
    A = ctx.upload(rp, ci, n)
    whole = ctx.multiply(A, A)
    st = ctx.stats()
    B_ = st["bins"]
    assert st["rows_per_bin"][B_ - 3] > 100 and st["rows_per_bin"][B_ - 2] > 10, st["rows_per_bin"]
    wrp, wci = whole.download()
    whole.free()
    cuts = [0, 1, 777, n // 3 + 5, n - 12345, n]
    for r0, r1 in zip(cuts[:-1], cuts[1:]):
        C = ctx.multiply(A, A, r0, r1)
        crp, cci = C.download()
        C.free()
        assert np.array_equal(crp, wrp[r0:r1 + 1] - wrp[r0]), (r0, r1)
        assert np.array_equal(cci, wci[wrp[r0]:wrp[r1]]), (r0, r1)
    erp, eci = O.spgemm_rows(rp, ci, rp, ci, n, 0, 4096)      # the heaviest rows of an R-MAT are its first
    assert np.array_equal(wrp[:4097], erp) and np.array_equal(wci[:wrp[4096]], eci)
    A.free()


def test_baseline_cfg4_rmat_scale24_shards(ctx):
    """BASELINE config 4 shape: R-MAT scale 24 (n = 16.7 M, five-bit levels = 4, nnz(C) = 5.45 G > 2^32),
    multiplied as 8 equal-work row shards like the 8-GPU run; every shard's rows sampled exactly,
    shard sizes add up, int64 row_ptr throughout."""
    import torch
    from bspgemm import dist as bdist
    dev = torch.device("cuda", 0)
    rp, ci, n = bspgemm.gen_rmat(24, 16, (0.30, 0.25, 0.25), seed=1)
    A = ctx.upload(rp, ci, n)
    prefix = ctx.row_work_prefix(A, A)
    bounds = bdist.shard_bounds(prefix, 8)
    total_nnz, total_F = 0, 0
    for p in range(8):
        r0, r1 = int(bounds[p]), int(bounds[p + 1])
        C = ctx.multiply(A, A, r0, r1)
        st = ctx.stats()
        crp, _ = C.download(col_idx=False)
        assert crp[0] == 0 and crp[-1] == C.nnz and np.all(np.diff(crp) >= 0)
        cols = bdist.device_tensor(C.col_idx_device, C.nnz, torch.int32, dev)
        for s0 in (0, (r1 - r0) // 2, r1 - r0 - 32):
            erp, eci = O.spgemm_rows(rp, ci, rp, ci, n, r0 + s0, r0 + s0 + 32)
            assert np.array_equal(crp[s0:s0 + 33] - crp[s0], erp)
            assert np.array_equal(cols[int(crp[s0]): int(crp[s0 + 32])].cpu().numpy(), eci)
        if p == 3:
            # one whole shard compared completely (cfg4 was only sampled): row_ptr and every column
            # index of these rows against the oracle's OpenMP restatement of SpGEMM_omp run on this row range
            _, cci = C.download()
            erp, eci = O.spgemm_omp(rp, ci, rp, ci, n, 4096, 0, row0=r0, rows=r1 - r0)
            assert np.array_equal(crp, erp)
            assert np.array_equal(cci, eci)
            del cci, erp, eci
        total_nnz += C.nnz
        total_F += st["products"]
        C.free()
    assert total_F == int(prefix[-1]) and total_nnz > 2**32
    work = np.diff(prefix[bounds])
    assert work.max() < 1.05 * work.mean()


def test_baseline_cfg5_powerlaw_full_size(ctx):
    """BASELINE config 5 at full size: power-law n = 2^20, mean degree 64 -- all capacity classes and
    ~170 K dense-window rows; well-formed, product count exact, hub and tail rows sampled exactly, then the
    whole product compared with the oracle."""
    import torch
    from bspgemm import dist as bdist
    rp, ci, n = bspgemm.gen_powerlaw(1 << 20, 64, seed=1)
    A = ctx.upload(rp, ci, n)
    C = ctx.multiply(A, A)
    st = ctx.stats()
    crp, _ = C.download(col_idx=False)
    assert crp[-1] == C.nnz and np.all(np.diff(crp) >= 0)
    assert st["products"] == O.count_products(rp, ci, rp) and st["rows_per_bin"][-1] > 1000
    cols = bdist.device_tensor(C.col_idx_device, C.nnz, torch.int32, torch.device("cuda", 0))
    deg = np.diff(rp)
    hub = int(np.argmax(deg))
    starts = [0, max(hub - 8, 0), n // 3, n - 64]
    _sampled_rows_exact(rp, ci, n, crp, cols, starts, block=16)
    del cols
    # everything (64 sampled rows were thin for the config whose point is the heavy-row kernels):
    # row_ptr and all 1.87 G column indices against the oracle's OpenMP run
    _, cci = C.download()
    C.free()
    erp, eci = O.spgemm_omp(rp, ci, rp, ci, n, 4096, 0)
    assert np.array_equal(crp, erp)
    assert np.array_equal(cci, eci)


def test_more_than_int32_output_nonzeros(ctx):
    """nnz(C) > 2^31-1 (what BASELINE configs 4 and 5 need and the reference's `int` counters cannot
    hold, final/SpGEMM_mpi_omp.c:20,111,177): int64 row_ptr end to end, sampled rows exact, and the
    int32 drop-in REFUSES instead of wrapping."""
    rp, ci, n = bspgemm.gen_uniform(1 << 22, 23, seed=5)
    A = ctx.upload(rp, ci, n)
    C = ctx.multiply(A, A)
    assert C.nnz > 2**31 - 1, C.nnz
    crp, _ = C.download(col_idx=False)
    assert crp[-1] == C.nnz and np.all(np.diff(crp) >= 0)
    st = ctx.stats()
    assert st["products"] == O.count_products(rp, ci, rp) and st["nnz_c"] == C.nnz
    # rows around the 2^31 boundary of the output and at both ends, straight from the device buffer
    import torch
    from bspgemm import dist as bdist
    dev = torch.device("cuda", 0)
    cols = bdist.device_tensor(C.col_idx_device, C.nnz, torch.int32, dev)
    r_mid = int(np.searchsorted(crp, 2**31)) - 2
    for r0 in (0, r_mid, n - 8):
        erp, eci = O.spgemm_rows(rp, ci, rp, ci, n, r0, r0 + 8)
        assert np.array_equal(crp[r0:r0 + 9] - crp[r0], erp)
        got = cols[int(crp[r0]): int(crp[r0 + 8])].cpu().numpy()
        assert np.array_equal(got, eci)
    C.free()
    with pytest.raises(bspgemm.BspgemmError) as e:
        bspgemm.SpGEMM_hip(ci, rp, n, ci, rp, n)
    assert e.value.status == 5          # BSPGEMM_ERR_OVERFLOW


def test_device_resident_closure(ctx):
    """bspgemm_closure: (A or I)^(2^k) to the fixpoint, products chained on the GPU"""
    sp = pytest.importorskip("scipy.sparse")
    n = 3000
    rng = np.random.default_rng(901)
    rows, cols = rng.integers(0, n, 3300), rng.integers(0, n, 3300)      # sparse digraph, long paths
    rp, ci = gen._csr_from_pairs(rows, cols, n)
    A = ctx.upload(rp, ci, n)
    T, products = ctx.closure(A)
    trp, tci = T.download()
    G = sp.csr_matrix((np.ones(ci.size, np.int8), ci, rp), shape=(n, n))
    from scipy.sparse.csgraph import floyd_warshall  # noqa: F401  (dense n=3000 is too big: use BFS reachability)
    from scipy.sparse.csgraph import breadth_first_order
    # reachability of a sample of sources by BFS == the closure's rows
    for src in (0, 17, 1234, n - 1):
        reach = np.sort(breadth_first_order(G, src, directed=True, return_predecessors=False))
        assert np.array_equal(tci[trp[src]:trp[src + 1]], reach), src
    assert 1 <= products <= 14
    # T is idempotent and exact: T*T == T through the oracle
    erp, eci = O.spgemm(trp.astype(np.int32), tci, trp.astype(np.int32), tci, n)
    assert_same(trp, tci, erp, eci)


def test_boolean_closure_is_idempotent(ctx):
    """(I + A)^k reaches a fixpoint T with T*T == T (the motivating use)"""
    n = 600
    rng = np.random.default_rng(701)
    rows = np.concatenate([np.arange(n), rng.integers(0, n, 500)])
    cols = np.concatenate([np.arange(n), rng.integers(0, n, 500)])
    rp, ci = gen._csr_from_pairs(rows, cols, n)
    for _ in range(12):
        A = ctx.upload(rp, ci, n)
        C = ctx.multiply(A, A)
        nrp, nci = C.download()
        same = np.array_equal(nrp, rp) and np.array_equal(nci, ci)
        rp, ci = nrp.astype(np.int32), nci
        if same:
            break
    assert same, "closure did not converge"
    erp, eci = O.spgemm(rp, ci, rp, ci, n)
    assert_same(rp.astype(np.int64), ci, erp, eci)
