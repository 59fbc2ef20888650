/*
 * csr_gen.c -- seeded synthetic boolean CSR matrices for the benchmark configs (host C, OpenMP).
 *
 * The reference's inputs were Matlab sprand patterns written by Matlab/write_spm.m:5-8
 * (`sprand(n,n,d/n)>0` -> mmwrite pattern); those files are not in the repository.  These
 * generators produce the BASELINE.json shapes directly in memory:
 *   uniform  : every row draws d columns i.i.d. uniform in [0,n)            (cfg 2)
 *   rmat     : R-MAT, 2^scale vertices, edge_factor*2^scale directed edges,
 *              no vertex permutation                                        (cfg 3, 4, stress)
 *   powerlaw : Pareto(alpha) out-degrees clipped to [1,max_degree] rescaled to the mean,
 *              DISTINCT columns drawn from an independent sample of the same law: the mean
 *              degree asked for is the mean degree the matrix has                  (cfg 5)
 * All: duplicates collapsed, col_idx ascending per row, int32 row_ptr.  Randomness is a
 * counter-based SplitMix64 keyed by (seed, element index), so the output does not depend on
 * the number of OpenMP threads.
 */
#include "../../include/bspgemm.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static inline uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static inline uint64_t rnd(uint64_t seed, uint64_t idx) { return splitmix64(splitmix64(seed) ^ (idx * 0xD1342543DE82EF95ull)); }
static inline double rnd01(uint64_t seed, uint64_t idx) { return (double)(rnd(seed, idx) >> 11) * (1.0 / 9007199254740992.0); }

static int cmp_int(const void *a, const void *b)
{
    const int x = *(const int *)a, y = *(const int *)b;
    return (x > y) - (x < y);
}

static void sort_row(int *a, long long n)
{
    if (n <= 32) {
        for (long long i = 1; i < n; i++) {
            int v = a[i];
            long long j = i - 1;
            while (j >= 0 && a[j] > v) { a[j + 1] = a[j]; j--; }
            a[j + 1] = v;
        }
    } else {
        qsort(a, (size_t)n, sizeof(int), cmp_int);
    }
}

/* rows given as (start[r], cols[...]) with duplicates, unsorted: sort + dedupe + compact.
 * `start` has n+1 int64 entries; consumes `cols`.                                         */
static bspgemm_status finish_csr(int n, long long *start, int *cols, int **row_ptr, int **col_idx)
{
    int *cnt = malloc(((size_t)n + 1) * sizeof(int));
    if (!cnt) { free(start); free(cols); return BSPGEMM_ERR_ALLOC; }
    #pragma omp parallel for schedule(dynamic, 1024)
    for (int r = 0; r < n; r++) {
        int *a = cols + start[r];
        const long long len = start[r + 1] - start[r];
        sort_row(a, len);
        long long k = 0;
        for (long long i = 0; i < len; i++)
            if (i == 0 || a[i] != a[i - 1]) a[k++] = a[i];
        cnt[r] = (int)k;
    }
    int *rp = malloc(((size_t)n + 1) * sizeof(int));
    if (!rp) { free(cnt); free(start); free(cols); return BSPGEMM_ERR_ALLOC; }
    long long total = 0;
    for (int r = 0; r < n; r++) { rp[r] = (int)total; total += cnt[r]; }
    if (total > 0x7fffffffll) { free(rp); free(cnt); free(start); free(cols); return BSPGEMM_ERR_OVERFLOW; }
    rp[n] = (int)total;
    int *ci = malloc((size_t)(total > 0 ? total : 1) * sizeof(int));
    if (!ci) { free(rp); free(cnt); free(start); free(cols); return BSPGEMM_ERR_ALLOC; }
    #pragma omp parallel for schedule(static)
    for (int r = 0; r < n; r++) memcpy(ci + rp[r], cols + start[r], (size_t)cnt[r] * sizeof(int));
    free(cnt); free(start); free(cols);
    *row_ptr = rp; *col_idx = ci;
    return BSPGEMM_OK;
}

bspgemm_status bspgemm_gen_uniform(int n, int d, uint64_t seed, int **row_ptr, int **col_idx)
{
    if (n <= 0 || d < 0 || !row_ptr || !col_idx) return BSPGEMM_ERR_INVALID;
    const long long m = (long long)n * d;
    long long *start = malloc(((size_t)n + 1) * sizeof(long long));
    int *cols = malloc((size_t)(m > 0 ? m : 1) * sizeof(int));
    if (!start || !cols) { free(start); free(cols); return BSPGEMM_ERR_ALLOC; }
    #pragma omp parallel for schedule(static)
    for (int r = 0; r <= n; r++) start[r] = (long long)r * d;
    #pragma omp parallel for schedule(static)
    for (long long e = 0; e < m; e++) cols[e] = (int)(rnd(seed, (uint64_t)e) % (uint64_t)n);
    return finish_csr(n, start, cols, row_ptr, col_idx);
}

/* edges -> CSR by counting sort on the row (shared by rmat and powerlaw) */
static bspgemm_status edges_to_csr(int n, long long m, int *er, int *ec, int **row_ptr, int **col_idx)
{
    long long *start = calloc((size_t)n + 2, sizeof(long long));
    int *cols = malloc((size_t)(m > 0 ? m : 1) * sizeof(int));
    if (!start || !cols) { free(start); free(cols); free(er); free(ec); return BSPGEMM_ERR_ALLOC; }
    for (long long e = 0; e < m; e++) start[er[e] + 1]++;
    for (int r = 0; r < n; r++) start[r + 1] += start[r];
    long long *cur = malloc(((size_t)n + 1) * sizeof(long long));
    if (!cur) { free(start); free(cols); free(er); free(ec); return BSPGEMM_ERR_ALLOC; }
    memcpy(cur, start, ((size_t)n + 1) * sizeof(long long));
    for (long long e = 0; e < m; e++) cols[cur[er[e]]++] = ec[e];
    free(cur); free(er); free(ec);
    return finish_csr(n, start, cols, row_ptr, col_idx);
}

bspgemm_status bspgemm_gen_rmat(int scale, int edge_factor, double a, double b, double c,
                                uint64_t seed, int **row_ptr, int **col_idx)
{
    if (scale < 1 || scale > 30 || edge_factor < 1 || !row_ptr || !col_idx) return BSPGEMM_ERR_INVALID;
    if (a < 0 || b < 0 || c < 0 || a + b + c > 1.0) return BSPGEMM_ERR_INVALID;
    const int n = 1 << scale;
    const long long m = (long long)edge_factor << scale;
    int *er = malloc((size_t)m * sizeof(int)), *ec = malloc((size_t)m * sizeof(int));
    if (!er || !ec) { free(er); free(ec); return BSPGEMM_ERR_ALLOC; }
    /* quadrant thresholds on a 32-bit draw: [0,a) top-left, [a,a+b) top-right, [a+b,a+b+c) bottom-left */
    const uint64_t ta = (uint64_t)(a * 4294967296.0), tb = (uint64_t)((a + b) * 4294967296.0),
                   tc = (uint64_t)((a + b + c) * 4294967296.0);
    #pragma omp parallel for schedule(static)
    for (long long e = 0; e < m; e++) {
        uint32_t r = 0, cl = 0;
        uint64_t bits = 0;
        for (int lvl = 0; lvl < scale; lvl++) {
            if ((lvl & 1) == 0) bits = rnd(seed, (uint64_t)e * 16 + (uint64_t)(lvl >> 1));
            const uint64_t u = (lvl & 1) ? (bits >> 32) : (bits & 0xffffffffull);
            const uint32_t down = u >= tb;                         /* bottom half */
            const uint32_t right = (u >= ta && u < tb) || u >= tc; /* right half */
            r = (r << 1) | down;
            cl = (cl << 1) | right;
        }
        er[e] = (int)r;
        ec[e] = (int)cl;
    }
    return edges_to_csr(n, m, er, ec, row_ptr, col_idx);
}

bspgemm_status bspgemm_gen_powerlaw(int n, int mean_degree, double alpha, int max_degree,
                                    uint64_t seed, int **row_ptr, int **col_idx)
{
    if (n <= 0 || mean_degree < 1 || alpha <= 1.0 || !row_ptr || !col_idx) return BSPGEMM_ERR_INVALID;
    if (max_degree <= 0) max_degree = n / 16 > 0 ? n / 16 : 1;
    double *w = malloc((size_t)n * sizeof(double));
    double *cdf = malloc(((size_t)n + 1) * sizeof(double));
    long long *start = malloc(((size_t)n + 1) * sizeof(long long));
    if (!w || !cdf || !start) { free(w); free(cdf); free(start); return BSPGEMM_ERR_ALLOC; }
    double wsum = 0;
    for (int i = 0; i < n; i++) {
        w[i] = pow(1.0 - rnd01(seed, (uint64_t)i), -1.0 / (alpha - 1.0));
        wsum += w[i];
    }
    /* scale s with sum_i clip(w_i*s, 1, max_degree) = n*mean_degree (bisection: the clipped sum
     * is monotone in s) -- "rescaled to the mean" AFTER clipping                            */
    double s_lo = 0.0, s_hi = (double)mean_degree * 64.0;
    for (int it = 0; it < 60; it++) {
        const double s = 0.5 * (s_lo + s_hi);
        double tot = 0;
        for (int i = 0; i < n; i++) {
            double dg = floor(w[i] * s);
            tot += dg < 1 ? 1 : (dg > max_degree ? max_degree : dg);
        }
        if (tot < (double)n * mean_degree) s_lo = s; else s_hi = s;
    }
    const double wmean = (double)mean_degree / s_hi;
    start[0] = 0;
    cdf[0] = 0;
    for (int i = 0; i < n; i++) {
        double dg = w[i] * mean_degree / wmean;
        long long deg = (long long)dg;
        if (deg < 1) deg = 1;
        if (deg > max_degree) deg = max_degree;
        start[i + 1] = start[i] + deg;
    }
    /* column popularity: an INDEPENDENT sample of the same Pareto law (a hub column need not be
     * a hub row), so F = sum |B_j| over A's nonzeros stays ~ nnz * mean degree */
    double csum = 0;
    for (int i = 0; i < n; i++) {
        w[i] = pow(1.0 - rnd01(seed ^ 0x5851F42D4C957F2Dull, (uint64_t)i), -1.0 / (alpha - 1.0));
        csum += w[i];
    }
    for (int i = 0; i < n; i++) cdf[i + 1] = cdf[i] + w[i] / csum;
    const long long m = start[n];
    int *cols = malloc((size_t)(m > 0 ? m : 1) * sizeof(int));
    if (!cols) { free(w); free(cdf); free(start); return BSPGEMM_ERR_ALLOC; }
    /* Row i gets deg_i DISTINCT columns: draws that repeat a column of the row are replaced (topped up in rounds:
     * sort, collapse, draw the shortfall again) so that the requested mean degree is what the matrix HAS after
     * duplicates collapse -- round 3's generator drew deg_i columns once and realised mean 52 for a requested 64
     * (hub rows lose a third of their draws to the popular columns).  Draw k of row i is keyed by (seed, i, k):
     * independent of the thread count.  A row stops after 64 rounds with what it has (only a row that asks for
     * nearly all reachable columns can get there).                                                             */
    #pragma omp parallel for schedule(dynamic, 256)
    for (int i = 0; i < n; i++) {
        int *a = cols + start[i];
        const long long deg = start[i + 1] - start[i];
        long long have = 0;
        uint64_t draw = 0;
        for (int round = 0; round < 64 && have < deg; round++) {
            for (long long k = have; k < deg; k++, draw++) {
                const double u = rnd01(seed ^ 0xA5A5A5A5ull, ((uint64_t)i << 28) ^ (draw * 0x9E3779B97F4A7C15ull)) * cdf[n];
                int lo = 0, hi = n;                          /* first c with cdf[c+1] > u */
                while (lo < hi) { const int mid = (lo + hi) >> 1; if (cdf[mid + 1] > u) hi = mid; else lo = mid + 1; }
                a[k] = lo < n ? lo : n - 1;
            }
            sort_row(a, deg);
            long long k = 0;
            for (long long q = 0; q < deg; q++)
                if (q == 0 || a[q] != a[q - 1]) a[k++] = a[q];
            have = k;
        }
        /* a row that gave up keeps its distinct columns; pad slots repeat the last one (finish_csr collapses them) */
        for (long long k = have; k < deg; k++) a[k] = a[have > 0 ? have - 1 : 0];
    }
    free(w); free(cdf);
    return finish_csr(n, start, cols, row_ptr, col_idx);
}
