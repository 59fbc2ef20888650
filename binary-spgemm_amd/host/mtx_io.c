/*
 * mtx_io.c -- Matrix Market side of the drop-in (host C, no GPU).
 *
 * bspgemm_readCOO mirrors readCOO (reference final/utils.c:47-81), including the two NIST
 * routines it calls (mm_read_banner final/mmio.c:96-179, mm_read_mtx_crd_size :189-217) and
 * the argument swap into coo2csc (utils.c:77, final/coo2csc.c:22-64): the pointer array is
 * built from the file's COLUMN index, the index array holds the file's ROW index, stable in
 * file order.  The result, used as CSR by every caller, is the transpose of the file's
 * matrix.  Same acceptance rules: banner must start with "%%MatrixMarket",
 * object "matrix", format coordinate|array, field real|complex|pattern|integer, symmetry
 * general|symmetric|hermitian|skew-symmetric; field and symmetry are then ignored for the
 * structure (symmetric files are NOT expanded), n = M.
 * Differences: failures are returned, not exit(1)'d (the CLI re-creates the exits, including which
 * of them print: a rejected banner is BSPGEMM_ERR_FORMAT, a rejected size line or entry
 * BSPGEMM_ERR_SIZE); the file is parsed from one buffer instead of nnz fscanf calls; for
 * real/integer/complex files the value tokens are skipped instead of being mis-read as
 * coordinates (:68 reads exactly two %u).
 * bspgemm_readCOO_ex adds, opt-in, what the reference leaves out: with
 * BSPGEMM_READ_EXPAND_SYMMETRIC the stored triangle of a symmetric / hermitian / skew-symmetric
 * file is mirrored (every off-diagonal entry (i,j) also gives (j,i), right behind it in file
 * order), so the CSR is the full pattern.  Without the flag the result is the reference's.
 */
#include "../../include/bspgemm.h"

#include <ctype.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static int ieq(const char *a, const char *b)   /* a lowered == b */
{
    for (; *a && *b; a++, b++)
        if (tolower((unsigned char)*a) != *b) return 0;
    return *a == 0 && *b == 0;
}

static const char *skip_ws(const char *p, const char *end)
{
    while (p < end && isspace((unsigned char)*p)) p++;
    return p;
}

/* parse one unsigned decimal token; returns NULL at end of buffer / on a non-number */
static const char *parse_u(const char *p, const char *end, unsigned long long *out)
{
    p = skip_ws(p, end);
    if (p < end && *p == '+') p++;
    if (p >= end || !isdigit((unsigned char)*p)) return NULL;
    unsigned long long v = 0;
    while (p < end && isdigit((unsigned char)*p)) v = v * 10 + (unsigned)(*p++ - '0');
    *out = v;
    return p;
}

static const char *skip_token(const char *p, const char *end)
{
    p = skip_ws(p, end);
    while (p < end && !isspace((unsigned char)*p)) p++;
    return p;
}

static const char *next_line(const char *p, const char *end)
{
    while (p < end && *p != '\n') p++;
    return p < end ? p + 1 : end;
}

bspgemm_status bspgemm_readCOO_ex(const char *path, unsigned flags, uint32_t **row, uint32_t **col,
                                  uint32_t *M, uint32_t *N, uint32_t *nnz)
{
    if (!path || !row || !col || !M || !N || !nnz) return BSPGEMM_ERR_INVALID;
    *row = *col = NULL;
    *M = *N = *nnz = 0;
    FILE *f = fopen(path, "rb");
    if (!f) return BSPGEMM_ERR_IO;                                   /* utils.c:54-55 */
    if (fseek(f, 0, SEEK_END) != 0) { fclose(f); return BSPGEMM_ERR_IO; }
    long sz = ftell(f);
    rewind(f);
    char *buf = malloc((size_t)sz + 1);
    if (!buf) { fclose(f); return BSPGEMM_ERR_ALLOC; }
    if (fread(buf, 1, (size_t)sz, f) != (size_t)sz) { free(buf); fclose(f); return BSPGEMM_ERR_IO; }
    fclose(f);
    buf[sz] = 0;
    const char *p = buf, *end = buf + sz;

    /* ---- banner (mmio.c:96-179) ---- */
    char banner[64], mtx[64], crd[64], dt[64], ss[64];
    {
        const char *eol = next_line(p, end);
        char line[1025];
        size_t len = (size_t)(eol - p) < sizeof line - 1 ? (size_t)(eol - p) : sizeof line - 1;
        memcpy(line, p, len);
        line[len] = 0;
        if (sscanf(line, "%63s %63s %63s %63s %63s", banner, mtx, crd, dt, ss) != 5) { free(buf); return BSPGEMM_ERR_FORMAT; }
        p = eol;
    }
    if (strncmp(banner, "%%MatrixMarket", 14) != 0) { free(buf); return BSPGEMM_ERR_FORMAT; }
    if (!ieq(mtx, "matrix")) { free(buf); return BSPGEMM_ERR_FORMAT; }
    if (!ieq(crd, "coordinate") && !ieq(crd, "array")) { free(buf); return BSPGEMM_ERR_FORMAT; }
    int extra;   /* value tokens per entry */
    if (ieq(dt, "pattern")) extra = 0;
    else if (ieq(dt, "real") || ieq(dt, "integer")) extra = 1;
    else if (ieq(dt, "complex")) extra = 2;
    else { free(buf); return BSPGEMM_ERR_FORMAT; }
    if (!ieq(ss, "general") && !ieq(ss, "symmetric") && !ieq(ss, "hermitian") && !ieq(ss, "skew-symmetric")) {
        free(buf);
        return BSPGEMM_ERR_FORMAT;
    }
    const int mirror = (flags & BSPGEMM_READ_EXPAND_SYMMETRIC) && !ieq(ss, "general");

    /* ---- size line (mmio.c:189-217): skip '%' lines, then three integers ---- */
    while (p < end && *p == '%') p = next_line(p, end);
    unsigned long long m = 0, n = 0, nz = 0;
    {
        const char *q = parse_u(p, end, &m);
        if (q) q = parse_u(q, end, &n);
        if (q) q = parse_u(q, end, &nz);
        if (!q) { free(buf); return BSPGEMM_ERR_SIZE; }                /* utils.c:60-61: exit(1), nothing printed */
        p = q;
    }
    if (m > 0x7fffffffull || n > 0x7fffffffull || nz > 0x7fffffffull) { free(buf); return BSPGEMM_ERR_SIZE; }
    if (mirror && 2 * nz > 0x7fffffffull) { free(buf); return BSPGEMM_ERR_SIZE; }

    /* ---- entries (utils.c:66-71) + coo2csc with swapped roles (utils.c:77) ---- */
    const size_t dim = (size_t)(m > n ? m : n);          /* the reference assumes square (coo2csc.c:18) */
    const size_t cap = (size_t)(nz ? nz : 1) * (mirror ? 2 : 1);
    uint32_t *I = malloc(cap * sizeof(uint32_t));
    uint32_t *J = malloc(cap * sizeof(uint32_t));
    uint32_t *rp = calloc(dim + 2, sizeof(uint32_t));
    uint32_t *ci = NULL;
    if (!I || !J || !rp) { free(I); free(J); free(rp); free(buf); return BSPGEMM_ERR_ALLOC; }
    const unsigned long long file_nz = nz;
    nz = 0;
    for (unsigned long long e = 0; e < file_nz; e++) {
        unsigned long long a = 1, b = 1;
        const char *q = parse_u(p, end, &a);
        if (q) q = parse_u(q, end, &b);
        if (!q || a < 1 || b < 1 || a > dim || b > dim) {
            free(I); free(J); free(rp); free(buf);
            return BSPGEMM_ERR_SIZE;
        }
        for (int x = 0; x < extra; x++) q = skip_token(q, end);
        p = q;
        I[nz] = (uint32_t)(a - 1);                                  /* 1-based -> 0-based */
        J[nz++] = (uint32_t)(b - 1);
        if (mirror && a != b) {                                     /* the other triangle */
            I[nz] = (uint32_t)(b - 1);
            J[nz++] = (uint32_t)(a - 1);
        }
    }
    free(buf);
    ci = malloc((size_t)(nz ? nz : 1) * sizeof(uint32_t));
    if (!ci) { free(I); free(J); free(rp); return BSPGEMM_ERR_ALLOC; }
    for (unsigned long long e = 0; e < nz; e++) rp[J[e] + 1]++;     /* coo2csc.c:37-39 */
    for (size_t i = 0; i < dim; i++) rp[i + 1] += rp[i];            /* :42-47 */
    uint32_t *cur = malloc((dim + 1) * sizeof(uint32_t));
    if (!cur) { free(I); free(J); free(rp); free(ci); return BSPGEMM_ERR_ALLOC; }
    memcpy(cur, rp, (dim + 1) * sizeof(uint32_t));
    for (unsigned long long e = 0; e < nz; e++) ci[cur[J[e]]++] = I[e];   /* :49-57, stable */
    free(cur); free(I); free(J);
    *row = rp; *col = ci;
    *M = (uint32_t)m; *N = (uint32_t)n; *nnz = (uint32_t)nz;
    return BSPGEMM_OK;
}

bspgemm_status bspgemm_readCOO(const char *path, uint32_t **row, uint32_t **col,
                               uint32_t *M, uint32_t *N, uint32_t *nnz)
{
    return bspgemm_readCOO_ex(path, 0u, row, col, M, N, nnz);
}

/* CSR (r, c) is written as the file entry "c+1 r+1": readCOO's transposition maps it back.
 * The file's matrix is therefore cols x rows (header "cols rows nnz"); square in every reference use. */
static bspgemm_status write_impl(const char *path, int rows, int cols, const int *rp32,
                                 const int64_t *rp64, const int *col_idx)
{
    if (!path || rows < 0 || cols < 0 || (!rp32 && !rp64)) return BSPGEMM_ERR_INVALID;
    FILE *f = fopen(path, "wb");
    if (!f) return BSPGEMM_ERR_IO;
    char *big = malloc((size_t)1 << 20);                    /* per call: the writer is re-entrant */
    if (big) setvbuf(f, big, _IOFBF, (size_t)1 << 20);
    const long long nnz = rp32 ? rp32[rows] - rp32[0] : rp64[rows] - rp64[0];
    fprintf(f, "%%%%MatrixMarket matrix coordinate pattern general\n%d %d %lld\n", cols, rows, nnz);
    for (int r = 0; r < rows; r++) {
        const long long b = rp32 ? rp32[r] : rp64[r], e = rp32 ? rp32[r + 1] : rp64[r + 1];
        for (long long k = b; k < e; k++) fprintf(f, "%d %d\n", col_idx[k] + 1, r + 1);
    }
    const int rc = fclose(f);
    free(big);
    return rc == 0 ? BSPGEMM_OK : BSPGEMM_ERR_IO;
}

bspgemm_status bspgemm_write_mtx(const char *path, int rows, int cols, const int *row_ptr, const int *col_idx)
{
    return write_impl(path, rows, cols, row_ptr, NULL, col_idx);
}

bspgemm_status bspgemm_write_result_mtx(const char *path, int rows, int cols, const int64_t *row_ptr,
                                        const int *col_idx)
{
    return write_impl(path, rows, cols, NULL, row_ptr, col_idx);
}

/* SpGEMM_valid, final/SpGEMM_mpi_omp_validity.c:290-302 */
int bspgemm_csr_equal(const int *Acol, const int *Arow, const int *Bcol, const int *Brow, int n)
{
    for (int i = 0; i <= n; i++) if (Arow[i] != Brow[i]) return 0;
    for (int i = 0; i < Arow[n]; i++) if (Acol[i] != Bcol[i]) return 0;
    return 1;
}

int bspgemm_csr_equal64(const int *Acol, const int64_t *Arow, const int *Bcol, const int64_t *Brow, int n)
{
    for (int i = 0; i <= n; i++) if (Arow[i] != Brow[i]) return 0;
    for (int64_t i = 0; i < Arow[n]; i++) if (Acol[i] != Bcol[i]) return 0;
    return 1;
}
