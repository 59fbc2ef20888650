"""One-off parity campaign beyond tests/: mid-size random products (R-MAT of several skews, power-law,
uniform, rectangular with hubs) against the CPU oracle, plain and masked.  usage: fuzz_big.py [cases [seed]]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "binary-spgemm_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import torch, bspgemm, gen
from oracle import oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 77
rng = np.random.default_rng(seed)
ctx = bspgemm.Context(0)
t0 = time.perf_counter()
for k in range(cases):
    kind = k % 6
    ctx.set_flow(("upper-bound", "exact", "auto")[(k // 5) % 3])
    # round 4: the per-operand paths in rotation as well (decided when an operand is first used as B: fresh operands per case)
    ctx.set_option("padded_rows", (0, 1, -1)[(k // 3) % 3])
    ctx.set_option("blocked_extents", (-1, 1, 0)[(k // 7) % 3])
    ctx.set_option("small_path", (-1, 0, 1)[(k // 11) % 3])
    if kind == 0:
        sc = int(rng.integers(12, 17)); abc = [(0.30, 0.25, 0.25), (0.45, 0.22, 0.22), (0.57, 0.19, 0.19)][k // 5 % 3]
        rp, ci, n = bspgemm.gen_rmat(sc, int(rng.integers(4, 24)), abc, seed=1000 * (seed - 76) + k)
        b_rp, b_ci, ncols = rp, ci, n
    elif kind == 1:
        n = int(rng.integers(5_000, 120_000))
        rp, ci, n = bspgemm.gen_powerlaw(n, int(rng.integers(4, 40)), seed=2000 * (seed - 76) + k)
        b_rp, b_ci, ncols = rp, ci, n
    elif kind == 5:
        # round 4: 2^18 < cols <= 2^20 with skew -- rows of 2-6 K products take the rank class (k_rank_rows), larger ones two to four windows
        sc = int(rng.integers(19, 21)); abc = [(0.45, 0.22, 0.22), (0.57, 0.19, 0.19), (0.50, 0.20, 0.20)][k // 6 % 3]
        rp, ci, n = bspgemm.gen_rmat(sc, int(rng.integers(2, 5)), abc, seed=5000 * (seed - 76) + k)
        b_rp, b_ci, ncols = rp, ci, n
    elif kind == 2:
        n = int(rng.integers(1_000, 300_000))
        rp, ci, n = bspgemm.gen_uniform(n, int(rng.integers(1, 30)), seed=3000 * (seed - 76) + k)
        b_rp, b_ci, ncols = rp, ci, n
    else:
        ar, inner = int(rng.integers(100, 20_000)), int(rng.integers(100, 20_000))
        ncols = int(rng.choice([5_000, 300_000, 5_000_000, 20_000_000, 600_000_000]))
        a_rows = np.concatenate([np.repeat(np.arange(ar), int(rng.integers(1, 20))), np.zeros(int(rng.integers(0, 5000)), np.int64)])
        a_cols = rng.integers(0, inner, size=a_rows.size)
        b_rows = np.concatenate([np.repeat(np.arange(inner), int(rng.integers(1, 60))), np.full(int(rng.integers(0, 50_000)), inner // 2)])
        span = ncols if kind == 3 else min(ncols, int(rng.integers(200, 100_000)))
        b_cols = rng.integers(0, span, size=b_rows.size)
        rp, ci = gen._csr_from_pairs(a_rows, a_cols, ar)
        b_rp, b_ci = gen._csr_from_pairs(b_rows, b_cols, inner)
        n = ar
    inner_cols = b_rp.size - 1
    erp, eci = O.spgemm(rp, ci, b_rp, b_ci, ncols)
    A = ctx.upload(rp, ci, inner_cols)
    B = A if b_rp is rp else ctx.upload(b_rp, b_ci, ncols)
    C = ctx.multiply(A, B)
    crp, cci = C.download()
    st = ctx.stats()
    ok = np.array_equal(crp, erp) and np.array_equal(cci, eci)
    mok = True
    if k % 3 == 0:                                   # masked by a random pattern + the matrix itself
        Fm = A if (b_rp is rp) else None
        if Fm is not None:
            M = ctx.multiply_masked(A, B, Fm)
            mrp, mci = M.download()
            frp, fci = O.spgemm_masked(rp, ci, b_rp, b_ci, ncols, rp, ci)
            mok = np.array_equal(mrp, frp) and np.array_equal(mci, fci)
            M.free()
    B_ = st["bins"]
    print("case %2d %-11s pad %d blk %d small %d kind %d rows %7d cols %9d products %.3g nnz %.3g heavy rows %d/%d/%d  %s %s" %
          (k, ("upper-bound", "exact", "auto")[(k // 5) % 3], st["padded_rows"], st["prepass_kernel"], st["small_path"], kind, n, ncols, st["products"], erp[-1],
           st["rows_per_bin"][B_ - 3], st["rows_per_bin"][B_ - 2], st["rows_per_bin"][B_ - 1],
           "OK" if ok else "MISMATCH", "" if mok else "MASKED MISMATCH"), flush=True)
    C.free()
    if B is not A:
        B.free()
    A.free()
    if not (ok and mok):
        sys.exit(1)
print("all %d cases bit-exact in %.0f s" % (cases, time.perf_counter() - t0))
