/*
 * par_copy.c -- host-side OpenMP helpers of the int32 drop-ins (no GPU code).  Internal to
 * libbspgemm.so (hidden visibility): not part of the C ABI.
 */
#include <stddef.h>
#include <string.h>

/* largest entry + 1 of an index array (the drop-ins derive B's row count from A.col_idx) */
__attribute__((visibility("hidden"))) int bspgemm_par_max_plus_one(const int *idx, long long n)
{
    int best = 0;
#pragma omp parallel for reduction(max : best) schedule(static)
    for (long long i = 0; i < n; i++)
        if (idx[i] >= best) best = idx[i] + 1;
    return best;
}

/* First touch of a fresh multi-GB destination by all threads at once: the kernel zeroes the pages
 * (2 MB each after MADV_HUGEPAGE) in parallel instead of inside the single-threaded copy path. */
__attribute__((visibility("hidden"))) void bspgemm_par_prefault(void *p, size_t bytes)
{
    const size_t step = 4096;
    const long long n = (long long)((bytes + step - 1) / step);
    volatile char *c = (volatile char *)p;
#pragma omp parallel for schedule(static)
    for (long long i = 0; i < n; i++) c[(size_t)i * step] = 0;
}

/* sum over rows [r0,r1) of min(F_i, cap) with F_i = sum_{j in A_i} |B_j|: how many entries the product of
 * those rows can have at most (|C_i| <= min(F_i, columns of B)).  One pass over A on all host threads: the
 * drop-ins size and fault in their malloc'ed result while the GPU is still busy.  Returns -1 if a column of
 * A is not a row of B (brows). */
__attribute__((visibility("hidden"))) long long bspgemm_par_output_bound(const int *Acol, const int *Arow, int r0, int r1,
                                                                        const int *Brow, int brows, long long cap)
{
    long long total = 0;
    int bad = 0;
#pragma omp parallel for reduction(+ : total) reduction(| : bad) schedule(dynamic, 4096)
    for (int i = r0; i < r1; i++) {
        long long f = 0;
        for (long long k = Arow[i]; k < Arow[i + 1]; k++) {
            const int j = Acol[k];
            if (j < 0 || j >= brows) { bad = 1; continue; }
            f += (long long)Brow[j + 1] - Brow[j];
        }
        total += f < cap ? f : cap;
    }
    return bad ? -1 : total;
}
