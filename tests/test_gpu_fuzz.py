"""Random parity campaign inside the driver's `-m gpu` run (tools/fuzz_big.py was builder-run
only).  30 seeded mid-size products -- R-MAT of three skews, power-law, uniform, rectangular A != B with
hubs and column counts up to 6e8 (five-bit levels 1..5) -- rotated over the flows, every third one also
masked; all compared bit for bit with the CPU oracle (oracle/bspgemm_oracle.c)."""
import numpy as np
import pytest

import bspgemm
import gen
from oracle import oracle as O

pytestmark = pytest.mark.gpu
FLOWS = ("upper-bound", "exact", "auto")


def _case(k, rng):
    kind = k % 5
    if kind == 0:
        sc = int(rng.integers(11, 15))
        abc = [(0.30, 0.25, 0.25), (0.45, 0.22, 0.22), (0.57, 0.19, 0.19)][k // 5 % 3]
        rp, ci, n = bspgemm.gen_rmat(sc, int(rng.integers(4, 20)), abc, seed=1000 + k)
        return rp, ci, rp, ci, n, n
    if kind == 1:
        rp, ci, n = bspgemm.gen_powerlaw(int(rng.integers(4_000, 40_000)), int(rng.integers(4, 24)), seed=2000 + k)
        return rp, ci, rp, ci, n, n
    if kind == 2:
        rp, ci, n = bspgemm.gen_uniform(int(rng.integers(1_000, 100_000)), int(rng.integers(1, 20)), seed=3000 + k)
        return rp, ci, rp, ci, n, n
    ar, inner = int(rng.integers(100, 8_000)), int(rng.integers(100, 8_000))
    ncols = int(rng.choice([5_000, 300_000, 5_000_000, 20_000_000, 600_000_000]))
    a_rows = np.concatenate([np.repeat(np.arange(ar), int(rng.integers(1, 12))), np.zeros(int(rng.integers(0, 3000)), np.int64)])
    a_cols = rng.integers(0, inner, size=a_rows.size)
    b_rows = np.concatenate([np.repeat(np.arange(inner), int(rng.integers(1, 40))), np.full(int(rng.integers(0, 20_000)), inner // 2)])
    span = ncols if kind == 3 else min(ncols, int(rng.integers(200, 100_000)))
    b_cols = rng.integers(0, span, size=b_rows.size)
    rp, ci = gen._csr_from_pairs(a_rows, a_cols, ar)
    b_rp, b_ci = gen._csr_from_pairs(b_rows, b_cols, inner)
    return rp, ci, b_rp, b_ci, ar, ncols


def test_random_campaign():
    rng = np.random.default_rng(77)
    ctx = bspgemm.Context(0)
    for k in range(30):
        flow = FLOWS[(k // 5 + k) % len(FLOWS)]
        ctx.set_flow(flow)
        rp, ci, b_rp, b_ci, n, ncols = _case(k, rng)
        inner = b_rp.size - 1
        erp, eci = O.spgemm(rp, ci, b_rp, b_ci, ncols)
        A = ctx.upload(rp, ci, inner)
        B = A if b_rp is rp else ctx.upload(b_rp, b_ci, ncols)
        C = ctx.multiply(A, B)
        crp, cci = C.download()
        C.free()
        assert np.array_equal(crp, erp) and np.array_equal(cci, eci), "case %d (%s flow, kind %d)" % (k, flow, k % 5)
        if k % 3 == 0 and b_rp is rp:                      # masked by the matrix itself
            M = ctx.multiply_masked(A, B, A)
            mrp, mci = M.download()
            M.free()
            frp, fci = O.spgemm_masked(rp, ci, b_rp, b_ci, ncols, rp, ci)
            assert np.array_equal(mrp, frp) and np.array_equal(mci, fci), "masked case %d" % k
        if B is not A:
            B.free()
        A.free()
    ctx.close()
