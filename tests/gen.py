"""Seeded numpy generators of small boolean CSR matrices for the tests and golden vectors.
(The product's own C generators live in binary-spgemm_amd/host/csr_gen.c and are tested
separately; these are independent so a generator bug cannot hide behind itself.)

All return int32 arrays: (row_ptr[n+1], col_idx[nnz], n) unless stated.
"""
import numpy as np


def _csr_from_pairs(rows, cols, n, dedup=True, sort=True):
    rows = np.asarray(rows, dtype=np.int64)
    cols = np.asarray(cols, dtype=np.int64)
    if dedup:
        key = np.unique(rows * (1 << 32) + cols)
        rows, cols = key >> 32, key & 0xFFFFFFFF
    elif sort:
        o = np.lexsort((cols, rows))
        rows, cols = rows[o], cols[o]
    else:
        o = np.argsort(rows, kind="stable")
        rows, cols = rows[o], cols[o]
    rp = np.zeros(n + 1, dtype=np.int64)
    np.add.at(rp, rows + 1, 1)
    rp = np.cumsum(rp)
    return rp.astype(np.int32), cols.astype(np.int32)


def uniform(n, d, seed):
    """each row draws d columns i.i.d. uniform in [0,n); duplicates collapsed (SURVEY 8d cfg 2)"""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n), d)
    cols = rng.integers(0, n, size=n * d)
    rp, ci = _csr_from_pairs(rows, cols, n)
    return rp, ci, n


def uniform_rect(nr, nc, d, seed):
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(nr), d)
    cols = rng.integers(0, nc, size=nr * d)
    return _csr_from_pairs(rows, cols, nr)


def rmat(scale, ef, abcd, seed):
    """R-MAT, no vertex permutation, directed, duplicates collapsed (SURVEY 8d cfg 3)"""
    rng = np.random.default_rng(seed)
    n = 1 << scale
    m = n * ef
    a, b, c, _ = abcd
    rows = np.zeros(m, dtype=np.int64)
    cols = np.zeros(m, dtype=np.int64)
    for _lvl in range(scale):
        r = rng.random(m)
        right = ((r >= a) & (r < a + b)) | (r >= a + b + c)
        down = r >= a + b
        rows = (rows << 1) | down
        cols = (cols << 1) | right
    rp, ci = _csr_from_pairs(rows, cols, n)
    return rp, ci, n


def with_special_rows(n, d, seed):
    """uniform, but a third of the rows empty, row 7 completely full, row n-1 with one entry"""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n), d)
    cols = rng.integers(0, n, size=n * d)
    keep = (rows % 3 != 0) & (rows != 7) & (rows != n - 1)
    rows, cols = rows[keep], cols[keep]
    rows = np.concatenate([rows, np.full(n, 7), [n - 1]])
    cols = np.concatenate([cols, np.arange(n), [n - 1]])
    rp, ci = _csr_from_pairs(rows, cols, n)
    return rp, ci, n


def dups_unsorted(n, d, seed):
    """rows keep duplicate entries and are NOT sorted (readCOO keeps duplicates)"""
    rng = np.random.default_rng(seed)
    rows = np.repeat(np.arange(n), d)
    cols = rng.integers(0, max(n // 8, 1), size=n * d)      # narrow range -> many duplicates
    rp, ci = _csr_from_pairs(rows, cols, n, dedup=False, sort=False)
    return rp, ci, n


def banded(n, half, seed):
    """band matrix with random holes: products collide heavily inside 64-column words"""
    rng = np.random.default_rng(seed)
    rows, cols = [], []
    for off in range(-half, half + 1):
        r = np.arange(max(0, -off), min(n, n - off))
        keep = rng.random(r.size) < 0.8
        rows.append(r[keep])
        cols.append(r[keep] + off)
    rp, ci = _csr_from_pairs(np.concatenate(rows), np.concatenate(cols), n)
    return rp, ci, n


def powerlaw(n, mean_deg, seed, alpha=2.1, max_deg=None):
    """out-degrees Pareto(alpha) clipped to [1,max_deg] rescaled to mean_deg; columns drawn from
    the same skewed distribution (SURVEY 8d cfg 5)"""
    rng = np.random.default_rng(seed)
    max_deg = max_deg or max(n // 16, 1)
    w = (1.0 - rng.random(n)) ** (-1.0 / (alpha - 1.0))
    deg = np.clip(w * mean_deg / w.mean(), 1, max_deg).astype(np.int64)
    rows = np.repeat(np.arange(n), deg)
    p = w / w.sum()
    cols = rng.choice(n, size=rows.size, p=p)
    rp, ci = _csr_from_pairs(rows, cols, n)
    return rp, ci, n
