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
