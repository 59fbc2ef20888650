/*
 * bspgemm_oracle.c -- CPU restatement of the reference's boolean SpGEMM hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product path (binary-spgemm_amd/, include/)
 * may link, load or call this file.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py use it, and only as the checker / the reported CPU baseline.
 *
 * Parity status: PINNED.  tests/test_oracle_pinned.py checks every function below against
 *   (1) the reference itself compiled from /root/reference into oracle/_ref (when present), and
 *   (2) the committed fixtures in tests/golden/ that were produced by that build
 *       (tests/golden/make_golden.py), including the reference's only committed input,
 *       Matlab/validity_test.mtx (nnz(A*A) = 12502).
 *
 * Every function cites the reference file:line it restates (paths relative to the
 * reference checkout).  The algorithm is the reference's: row-wise Gustavson with a dense
 * per-thread flag array, unsorted append, per-row ascending sort, sparse flag reset.
 * Deliberate differences, all outside the result:
 *   - C.row_ptr is int64 (the reference overflows int above 2^31-1 output nonzeros,
 *     final/SpGEMM_mpi_omp.c:20,111,177).
 *   - the per-row sort is an introsort-style quicksort (median-of-3 + insertion sort) instead
 *     of the reference's Lomuto/last-pivot quickSort (final/utils.c:130-173), which is
 *     quadratic on already sorted rows.  The sorted order is the same.
 *   - remainder rows (An % tBlock) are computed as a final short slice instead of being
 *     silently dropped (final/SpGEMM_mpi_omp.c:77).
 */
#include <stdint.h>
#include <stdlib.h>
#include <stdio.h>
#include <string.h>
#include <ctype.h>
#include <stdbool.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ------------------------------------------------------------------------------------ */
/* ascending sort of one C row -- stands in for quickSort, final/utils.c:159-173         */
static void ins_sort(int *a, int64_t lo, int64_t hi)
{
    for (int64_t i = lo + 1; i <= hi; i++) {
        int v = a[i];
        int64_t j = i - 1;
        while (j >= lo && a[j] > v) { a[j + 1] = a[j]; j--; }
        a[j + 1] = v;
    }
}

static void sort_asc(int *a, int64_t lo, int64_t hi)
{
    while (hi - lo > 24) {
        int64_t mid = lo + (hi - lo) / 2;
        int x = a[lo], y = a[mid], z = a[hi];
        int pivot = (x < y) ? ((y < z) ? y : (x < z ? z : x)) : ((x < z) ? x : (y < z ? z : y));
        int64_t i = lo, j = hi;
        while (i <= j) {
            while (a[i] < pivot) i++;
            while (a[j] > pivot) j--;
            if (i <= j) { int t = a[i]; a[i] = a[j]; a[j] = t; i++; j--; }
        }
        /* recurse on the smaller half, loop on the larger */
        if (j - lo < hi - i) { if (lo < j) sort_asc(a, lo, j); lo = i; }
        else                 { if (i < hi) sort_asc(a, i, hi); hi = j; }
    }
    if (hi > lo) ins_sort(a, lo, hi);
}

/* ------------------------------------------------------------------------------------ */
/* Restates SpGEMM_bigslice, final/SpGEMM_mpi_omp.c:15-58 (twin: Matlab/inc/BSpGEMM.c:9-47).
 * Rows [start_row,end_row) of C = A*B over the OR/AND semiring.
 *   Arow holds ABSOLUTE offsets into Acol (also for interior pointers, :171).
 *   Crow[0..rows] is slice-local (starts at 0, :26,:54); *Ccol grows like :28-31.
 * Returns nnz of the slice, or -1 on allocation failure.                                  */
int64_t oracle_bigslice(const int *Acol, const int *Arow,
                        const int *Bcol, const int *Brow, int Bm,
                        int **Ccol, int64_t *Crow, int64_t *Csize,
                        int start_row, int end_row)
{
    int64_t nnzcum = 0;
    bool *xb = calloc((size_t)(Bm > 0 ? Bm : 1), sizeof(bool));       /* :21 */
    if (!xb) return -1;
    int64_t ip = 0;

    for (int i = start_row; i < end_row; i++) {                       /* :24 */
        int64_t nnzpv = nnzcum;
        Crow[ip++] = nnzcum;                                          /* :26 */
        if (nnzcum + Bm > *Csize) {                                   /* :28-31 */
            int64_t grow = (*Csize / 4 > Bm) ? *Csize / 4 : Bm;
            *Csize += grow;
            int *p = realloc(*Ccol, (size_t)(*Csize) * sizeof(int));
            if (!p) { free(xb); return -1; }
            *Ccol = p;
        }
        for (int jj = Arow[i]; jj < Arow[i + 1]; jj++) {              /* :33 */
            int j = Acol[jj];
            for (int kp = Brow[j]; kp < Brow[j + 1]; kp++) {          /* :36 */
                int k = Bcol[kp];
                if (!xb[k]) {                                         /* :38-42 */
                    xb[k] = true;
                    (*Ccol)[nnzcum++] = k;
                }
            }
        }
        if (nnzcum > nnzpv) {                                         /* :46-51 */
            sort_asc(*Ccol, nnzpv, nnzcum - 1);
            for (int64_t p = nnzpv; p < nnzcum; p++) xb[(*Ccol)[p]] = false;
        }
    }
    Crow[ip] = nnzcum;                                                /* :54 */
    free(xb);
    return nnzcum;
}

/* Number of flag probes ("products") F = sum over A-nonzeros (i,j) of |B_j| for rows
 * [start,end).  Not a reference function: the quantity SURVEY.md 8(d) prices bytes with.   */
int64_t oracle_count_products(const int *Acol, const int *Arow, const int *Brow,
                              int start_row, int end_row)
{
    int64_t F = 0;
    for (int i = start_row; i < end_row; i++)
        for (int jj = Arow[i]; jj < Arow[i + 1]; jj++)
            F += Brow[Acol[jj] + 1] - Brow[Acol[jj]];
    return F;
}

/* ------------------------------------------------------------------------------------ */
/* Restates SpGEMM_omp, final/SpGEMM_mpi_omp.c:71-143: An rows cut into slices of tBlock
 * rows, each slice = one oracle_bigslice under omp-for schedule(static), then concat of the
 * Ccol slices (:119-127) and serial rebase of Crow (:134-141).
 * Arow may be an interior pointer (rows are 0..An-1 relative to it, values absolute).
 * *Ccol is malloc'd here (exact size, :115); Crow has An+1 entries, caller-owned.
 * Returns total nnz or -1.                                                                */
int64_t oracle_spgemm_omp(const int *Acol, const int *Arow, int An,
                          const int *Bcol, const int *Brow, int Bm,
                          int **Ccol, int64_t *Crow, int tBlock, int threads)
{
    if (tBlock <= 0) tBlock = An > 0 ? An : 1;
    int slices = (An + tBlock - 1) / tBlock;                          /* :77 (+ remainder) */
    int **cc = calloc((size_t)(slices > 0 ? slices : 1), sizeof(int *));
    int64_t **cr = calloc((size_t)(slices > 0 ? slices : 1), sizeof(int64_t *));
    int64_t *csz = calloc((size_t)(slices > 0 ? slices : 1), sizeof(int64_t));
    int64_t *cnnz = calloc((size_t)(slices > 0 ? slices : 1), sizeof(int64_t));
    int bad = 0;
    for (int s = 0; s < slices; s++) {                                /* :88-92 */
        csz[s] = Bm > 0 ? Bm : 1;
        cc[s] = malloc((size_t)csz[s] * sizeof(int));
        cr[s] = malloc((size_t)(tBlock + 1) * sizeof(int64_t));
        if (!cc[s] || !cr[s]) bad = 1;
    }
    if (!bad) {
#ifdef _OPENMP
        if (threads > 0) omp_set_num_threads(threads);
#endif
        #pragma omp parallel for schedule(static)                     /* :95-108 */
        for (int s = 0; s < slices; s++) {
            int r0 = s * tBlock;
            int r1 = r0 + tBlock < An ? r0 + tBlock : An;
            cnnz[s] = oracle_bigslice(Acol, Arow, Bcol, Brow, Bm, &cc[s], cr[s], &csz[s], r0, r1);
        }
        for (int s = 0; s < slices; s++) if (cnnz[s] < 0) bad = 1;
    }
    int64_t nnz = 0;
    if (!bad) {
        for (int s = 0; s < slices; s++) nnz += cnnz[s];              /* :110-113 */
        *Ccol = malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int));    /* :115 */
        if (!*Ccol) bad = 1;
    }
    if (!bad) {
        Crow[0] = 0;                                                  /* :118 */
        int64_t base = 0;
        for (int s = 0; s < slices; s++) {                            /* :120-141 fused */
            int r0 = s * tBlock;
            int r1 = r0 + tBlock < An ? r0 + tBlock : An;
            memcpy(*Ccol + base, cc[s], (size_t)cnnz[s] * sizeof(int));
            for (int j = 1; j <= r1 - r0; j++) Crow[r0 + j] = cr[s][j] + base;
            base += cnnz[s];
        }
    }
    for (int s = 0; s < slices; s++) { free(cc[s]); free(cr[s]); }
    free(cc); free(cr); free(csz); free(cnnz);
    return bad ? -1 : nnz;
}

/* Whole-matrix serial product with exact-size output: restates SpGEMM_mat,
 * Matlab/inc/BSpGEMM.c:9-47 (flag array sized by Bm here, not An -- SURVEY.md 9.8).       */
int64_t oracle_spgemm(const int *Acol, const int *Arow, int An,
                      const int *Bcol, const int *Brow, int Bm,
                      int **Ccol, int64_t *Crow)
{
    int64_t csize = Bm > 0 ? Bm : 1;
    *Ccol = malloc((size_t)csize * sizeof(int));
    if (!*Ccol) return -1;
    return oracle_bigslice(Acol, Arow, Bcol, Brow, Bm, Ccol, Crow, &csize, 0, An);
}

/* ------------------------------------------------------------------------------------ */
/* Restates SpGEMM_masked, final/SpGEMM_mpi_omp.c:232-288:  C = F .* (A*B).
 * Flags start true (:240-241); the row's mask columns are set false (:253-255) so only
 * masked-in columns can be appended; after the row they are set back to true (:279-281).
 * The reference sizes xb by An (square); here by Bm.                                      */
int64_t oracle_spgemm_masked(const int *Acol, const int *Arow, int An,
                             const int *Bcol, const int *Brow, int Bm,
                             const int *Fcol, const int *Frow,
                             int **Ccol, int64_t *Crow)
{
    int64_t csize = Bm > 0 ? Bm : 1, nnzcum = 0;
    *Ccol = malloc((size_t)csize * sizeof(int));
    bool *xb = malloc((size_t)(Bm > 0 ? Bm : 1) * sizeof(bool));
    if (!*Ccol || !xb) { free(xb); return -1; }
    for (int i = 0; i < Bm; i++) xb[i] = true;                        /* :241 */
    for (int i = 0; i < An; i++) {
        int64_t nnzpv = nnzcum;
        Crow[i] = nnzcum;                                             /* :245 */
        if (nnzcum + Bm > csize) {                                    /* :246-249 */
            csize += (csize / 4 > Bm) ? csize / 4 : Bm;
            int *p = realloc(*Ccol, (size_t)csize * sizeof(int));
            if (!p) { free(xb); return -1; }
            *Ccol = p;
        }
        for (int jj = Frow[i]; jj < Frow[i + 1]; jj++) xb[Fcol[jj]] = false;   /* :253 */
        for (int jj = Arow[i]; jj < Arow[i + 1]; jj++) {                       /* :259 */
            int j = Acol[jj];
            for (int kp = Brow[j]; kp < Brow[j + 1]; kp++) {
                int k = Bcol[kp];
                if (!xb[k]) { xb[k] = true; (*Ccol)[nnzcum++] = k; }
            }
        }
        if (nnzcum > nnzpv) {                                         /* :272-277 */
            sort_asc(*Ccol, nnzpv, nnzcum - 1);
            for (int64_t p = nnzpv; p < nnzcum; p++) xb[(*Ccol)[p]] = false;
        }
        for (int jj = Frow[i]; jj < Frow[i + 1]; jj++) xb[Fcol[jj]] = true;    /* :279 */
    }
    Crow[An] = nnzcum;                                                /* :284 */
    free(xb);
    return nnzcum;
}

/* ------------------------------------------------------------------------------------ */
/* Restates SpGEMM_valid, final/SpGEMM_mpi_omp_validity.c:290-302: exact CSR equality
 * (all n+1 row pointers, then all col indices).  1 = equal.  Second CSR may carry int64
 * row pointers (native) -- both flavours offered.                                         */
int oracle_csr_equal64(const int *Acol, const int64_t *Arow,
                       const int *Bcol, const int64_t *Brow, int n)
{
    for (int i = 0; i <= n; i++) if (Arow[i] != Brow[i]) return 0;
    for (int64_t i = 0; i < Arow[n]; i++) if (Acol[i] != Bcol[i]) return 0;
    return 1;
}

int oracle_csr_equal32(const int *Acol, const int *Arow,
                       const int *Bcol, const int *Brow, int n)
{
    for (int i = 0; i <= n; i++) if (Arow[i] != Brow[i]) return 0;
    for (int i = 0; i < Arow[n]; i++) if (Acol[i] != Bcol[i]) return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------ */
/* Restates readCOO, final/utils.c:47-81, with the two mmio.c routines it uses
 * (mm_read_banner final/mmio.c:96-179, mm_read_mtx_crd_size :189-217) and coo2csc
 * (final/coo2csc.c:22-64) called with swapped roles (utils.c:77): the pointer array counts
 * the file's COLUMN index J, the index array receives the file's ROW index I, stable in
 * file order -- i.e. the returned "CSR" is the transpose of the file's matrix.
 * Returns 0, or: 1 open failed (reference: silent exit(1), utils.c:54), 2 bad banner
 * (reference prints "Could not process Matrix Market banner." and exit(1), :56-59),
 * 3 bad size line (:60-61), 4 allocation.  Entries are read as two unsigned tokens each
 * (:68), whatever the banner's field says.                                                */
static int ci_eq(const char *a, const char *b)
{
    for (; *a && *b; a++, b++) if (tolower((unsigned char)*a) != *b) return 0;
    return *a == 0 && *b == 0;
}

int oracle_readCOO(const char *path, int **row_ptr, int **col_idx, int *M, int *N, int *nnz)
{
    FILE *f = fopen(path, "r");
    if (!f) return 1;
    char line[1025];
    char banner[64], mtx[64], crd[64], dt[64], ss[64];
    if (!fgets(line, sizeof line, f)) { fclose(f); return 2; }
    if (sscanf(line, "%63s %63s %63s %63s %63s", banner, mtx, crd, dt, ss) != 5) { fclose(f); return 2; }
    if (strncmp(banner, "%%MatrixMarket", 14) != 0) { fclose(f); return 2; }          /* mmio.c:122 */
    if (!ci_eq(mtx, "matrix")) { fclose(f); return 2; }
    if (!ci_eq(crd, "coordinate") && !ci_eq(crd, "array")) { fclose(f); return 2; }
    if (!ci_eq(dt, "real") && !ci_eq(dt, "complex") && !ci_eq(dt, "pattern") && !ci_eq(dt, "integer")) { fclose(f); return 2; }
    if (!ci_eq(ss, "general") && !ci_eq(ss, "symmetric") && !ci_eq(ss, "hermitian") && !ci_eq(ss, "skew-symmetric")) { fclose(f); return 2; }
    do {                                                                               /* mmio.c:198-202 */
        if (!fgets(line, sizeof line, f)) { fclose(f); return 3; }
    } while (line[0] == '%');
    int m = 0, n = 0, nz = 0;
    if (sscanf(line, "%d %d %d", &m, &n, &nz) != 3) {
        int got;
        do { got = fscanf(f, "%d %d %d", &m, &n, &nz); if (got == EOF) { fclose(f); return 3; } } while (got != 3);
    }
    uint32_t *I = malloc((size_t)(nz > 0 ? nz : 1) * sizeof(uint32_t));
    uint32_t *J = malloc((size_t)(nz > 0 ? nz : 1) * sizeof(uint32_t));
    int *rp = calloc((size_t)(m > n ? m : n) + 2, sizeof(int));   /* reference assumes square, coo2csc.c:18 */
    int *ci = malloc((size_t)(nz > 0 ? nz : 1) * sizeof(int));
    if (!I || !J || !rp || !ci) { free(I); free(J); free(rp); free(ci); fclose(f); return 4; }
    for (int i = 0; i < nz; i++) {                                                     /* utils.c:66-71 */
        if (fscanf(f, "%u %u\n", &I[i], &J[i]) != 2) { I[i] = 1; J[i] = 1; }
        I[i]--; J[i]--;
    }
    fclose(f);
    /* coo2csc(*col,*row,I,J,nnz,M,0): count J, place I (coo2csc.c:37-56) */
    for (int l = 0; l < nz; l++) rp[J[l] + 1]++;
    for (int i = 0; i < (m > n ? m : n); i++) rp[i + 1] += rp[i];
    int *cur = malloc(((size_t)(m > n ? m : n) + 1) * sizeof(int));
    if (!cur) { free(I); free(J); free(rp); free(ci); return 4; }
    memcpy(cur, rp, ((size_t)(m > n ? m : n) + 1) * sizeof(int));
    for (int l = 0; l < nz; l++) ci[cur[J[l]]++] = (int)I[l];
    free(cur); free(I); free(J);
    *row_ptr = rp; *col_idx = ci; *M = m; *N = n; *nnz = nz;
    return 0;
}

void oracle_free(void *p) { free(p); }
