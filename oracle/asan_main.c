/*
 * asan_main.c -- sanitizer driver of the host C (the reference ships no sanitizer build).
 *
 * TEST INFRASTRUCTURE ONLY (built by oracle/Makefile as asan/host_asan with
 * -fsanitize=address,undefined; run by tests/test_host_c.py, CPU only).  It links the PRODUCT's host C
 * (binary-spgemm_amd/host/mtx_io.c, csr_gen.c, par_copy.c) next to the checker's restatement
 * (bspgemm_oracle.c) and walks every entry point of both on small inputs, including the loaders' error
 * paths on malformed files.  Any heap / stack / UB report aborts the program; exit 0 = clean.
 *
 *   usage: host_asan <scratch directory> <path to tests/golden/validity_test.mtx>
 */
#include "bspgemm.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* the restatement (bspgemm_oracle.c) */
int64_t oracle_bigslice(const int *, const int *, const int *, const int *, int, int **, int64_t *, int64_t *, int, int);
int64_t oracle_count_products(const int *, const int *, const int *, int, int);
int64_t oracle_spgemm_omp(const int *, const int *, int, const int *, const int *, int, int **, int64_t *, int, int);
int64_t oracle_spgemm(const int *, const int *, int, const int *, const int *, int, int **, int64_t *);
int64_t oracle_spgemm_masked(const int *, const int *, int, const int *, const int *, int, const int *, const int *, int **, int64_t *);
int oracle_csr_equal64(const int *, const int64_t *, const int *, const int64_t *, int);
int oracle_csr_equal32(const int *, const int *, const int *, const int *, int);
int oracle_readCOO(const char *, int **, int **, int *, int *, int *);
/* hidden helpers of the drop-ins (par_copy.c) */
int bspgemm_par_max_plus_one(const int *idx, long long n);
void bspgemm_par_prefault(void *p, size_t bytes);
long long bspgemm_par_output_bound(const int *Acol, const int *Arow, int r0, int r1, const int *Brow, int brows, long long cap);

#define CHECK(cond)                                                                 \
    do {                                                                            \
        if (!(cond)) {                                                              \
            fprintf(stderr, "host_asan: %s failed (line %d)\n", #cond, __LINE__);   \
            exit(2);                                                                \
        }                                                                           \
    } while (0)

static void write_text(const char *path, const char *text)
{
    FILE *f = fopen(path, "w");
    CHECK(f != NULL);
    fputs(text, f);
    fclose(f);
}

/* product of A*A three ways through the restatement; returns nnz */
static int64_t products_agree(const int *rp, const int *ci, int n)
{
    int *c1 = NULL, *c2 = NULL, *c3 = NULL;
    int64_t *r1 = malloc(((size_t)n + 1) * sizeof(int64_t)), *r2 = malloc(((size_t)n + 1) * sizeof(int64_t));
    int64_t *r3 = malloc(((size_t)n + 1) * sizeof(int64_t));
    CHECK(r1 && r2 && r3);
    const int64_t z1 = oracle_spgemm(ci, rp, n, ci, rp, n, &c1, r1);
    const int64_t z2 = oracle_spgemm_omp(ci, rp, n, ci, rp, n, &c2, r2, n / 7 + 1, 3);   /* ragged last slice */
    int64_t csize = 4;
    c3 = malloc((size_t)csize * sizeof(int));
    const int64_t z3 = oracle_bigslice(ci, rp, ci, rp, n, &c3, r3, &csize, 0, n);         /* grows by realloc */
    CHECK(z1 >= 0 && z1 == z2 && z1 == z3);
    CHECK(oracle_csr_equal64(c1, r1, c2, r2, n) && oracle_csr_equal64(c1, r1, c3, r3, n));
    CHECK(bspgemm_csr_equal64(c1, r1, c2, r2, n) == 1);
    CHECK(oracle_count_products(ci, rp, rp, 0, n) >= z1);
    /* masked by A itself: a subset of the product */
    int *cm = NULL;
    const int64_t zm = oracle_spgemm_masked(ci, rp, n, ci, rp, n, ci, rp, &cm, r2);
    CHECK(zm >= 0 && zm <= z1 && zm <= rp[n]);
    /* the drop-ins' host helpers */
    CHECK(bspgemm_par_max_plus_one(ci, rp[n]) <= n);
    CHECK(bspgemm_par_output_bound(ci, rp, 0, n, rp, n, n) >= z1);
    free(c1); free(c2); free(c3); free(cm); free(r1); free(r2); free(r3);
    return z1;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: host_asan <scratch dir> <validity_test.mtx>\n"); return 1; }
    char path[4096];
    /* ---- generators -> restatement -> writers -> loaders ------------------------------------------- */
    int *rp = NULL, *ci = NULL;
    CHECK(bspgemm_gen_uniform(700, 5, 11, &rp, &ci) == BSPGEMM_OK);
    const int64_t nnz_u = products_agree(rp, ci, 700);
    snprintf(path, sizeof path, "%s/u.mtx", argv[1]);
    CHECK(bspgemm_write_mtx(path, 700, 700, rp, ci) == BSPGEMM_OK);
    uint32_t *lr = NULL, *lc = NULL, M = 0, N = 0, nz = 0;
    CHECK(bspgemm_readCOO(path, &lr, &lc, &M, &N, &nz) == BSPGEMM_OK);
    CHECK(M == 700 && N == 700 && nz == (uint32_t)rp[700]);
    CHECK(bspgemm_csr_equal((const int *)lc, (const int *)lr, ci, rp, 700) == 1);   /* the writer undoes the loader's transpose */
    int *orp = NULL, *oci = NULL, oM = 0, oN = 0, onz = 0;
    CHECK(oracle_readCOO(path, &orp, &oci, &oM, &oN, &onz) == 0);
    CHECK(oM == 700 && onz == rp[700] && oracle_csr_equal32(oci, orp, ci, rp, 700));
    free(orp); free(oci); free(lr); free(lc);
    free(rp); free(ci);

    CHECK(bspgemm_gen_rmat(9, 8, 0.57, 0.19, 0.19, 3, &rp, &ci) == BSPGEMM_OK);
    products_agree(rp, ci, 512);
    /* the product written in the FILE's orientation with int64 row_ptr, read back */
    {
        int *cc = NULL;
        int64_t *cr = malloc(513 * sizeof(int64_t));
        CHECK(cr && oracle_spgemm(ci, rp, 512, ci, rp, 512, &cc, cr) >= 0);
        snprintf(path, sizeof path, "%s/c.mtx", argv[1]);
        CHECK(bspgemm_write_result_mtx(path, 512, 512, cr, cc) == BSPGEMM_OK);
        CHECK(bspgemm_readCOO(path, &lr, &lc, &M, &N, &nz) == BSPGEMM_OK && nz == (uint32_t)cr[512]);
        free(lr); free(lc); free(cc); free(cr);
    }
    free(rp); free(ci);

    CHECK(bspgemm_gen_powerlaw(2000, 12, 2.1, 0, 5, &rp, &ci) == BSPGEMM_OK);
    CHECK(rp[2000] == 2000 * 12);                           /* the mean degree asked for is the one realised */
    products_agree(rp, ci, 2000);
    free(rp); free(ci);
    CHECK(bspgemm_gen_uniform(0, 5, 1, &rp, &ci) == BSPGEMM_ERR_INVALID);
    CHECK(bspgemm_gen_rmat(40, 1, 0.3, 0.3, 0.3, 1, &rp, &ci) == BSPGEMM_ERR_INVALID);

    /* ---- the reference's fixture through both loaders ------------------------------------------------ */
    CHECK(bspgemm_readCOO(argv[2], &lr, &lc, &M, &N, &nz) == BSPGEMM_OK && M == 50000 && nz == 25000);
    CHECK(oracle_readCOO(argv[2], &orp, &oci, &oM, &oN, &onz) == 0 && onz == 25000);
    CHECK(oracle_csr_equal32(oci, orp, (const int *)lc, (const int *)lr, 50000));
    CHECK(products_agree(orp, oci, 50000) == 12502);        /* BASELINE config 1 */
    free(orp); free(oci); free(lr); free(lc);

    /* ---- symmetric expansion and value tokens -------------------------------------------------------- */
    snprintf(path, sizeof path, "%s/s.mtx", argv[1]);
    write_text(path, "%%MatrixMarket matrix coordinate real symmetric\n% comment\n4 4 4\n1 1 1.5\n3 1 2.0\n4 2 -1e3\n4 4 7\n");
    CHECK(bspgemm_readCOO_ex(path, BSPGEMM_READ_EXPAND_SYMMETRIC, &lr, &lc, &M, &N, &nz) == BSPGEMM_OK && nz == 6);
    free(lr); free(lc);
    CHECK(bspgemm_readCOO(path, &lr, &lc, &M, &N, &nz) == BSPGEMM_OK && nz == 4);
    free(lr); free(lc);

    /* ---- malformed files: every rejection path, nothing leaked or overrun ---------------------------- */
    static const char *bad[] = {
        "",                                                                          /* empty file */
        "%MatrixMarket matrix coordinate pattern general\n2 2 1\n1 1\n",             /* one % : banner rejected */
        "%%MatrixMarket matrix array real general\n2 2\n1.0\n",                      /* not coordinate */
        "%%MatrixMarket matrix coordinate pattern general\nthree by three\n",        /* bad size line */
        "%%MatrixMarket matrix coordinate pattern general\n3 3 4\n1 1\n2 2\n",       /* fewer entries than announced */
        "%%MatrixMarket matrix coordinate pattern general\n3 3 2\n1 1\n9 9\n",       /* index out of range */
        "%%MatrixMarket matrix coordinate pattern general\n3 3 2\n1 1\n0 2\n",       /* zero index */
        "%%MatrixMarket matrix coordinate pattern general\n3 3 2\n1 x\n2 2\n",       /* junk token */
        "%%MatrixMarket matrix coordinate pattern general\n99999999999 3 2\n1 1\n",  /* size overflows */
        "%%MatrixMarket matrix coordinate pattern general\n3 3 2\n1 1\n2",           /* truncated last line */
    };
    for (size_t k = 0; k < sizeof bad / sizeof bad[0]; k++) {
        snprintf(path, sizeof path, "%s/bad%zu.mtx", argv[1], k);
        write_text(path, bad[k]);
        lr = lc = NULL;
        const bspgemm_status st = bspgemm_readCOO_ex(path, k & 1 ? BSPGEMM_READ_EXPAND_SYMMETRIC : 0u, &lr, &lc, &M, &N, &nz);
        if (st == BSPGEMM_OK) { free(lr); free(lc); }        /* (a loader may accept what the reference's fscanf accepts) */
        else CHECK(st == BSPGEMM_ERR_IO || st == BSPGEMM_ERR_FORMAT || st == BSPGEMM_ERR_SIZE || st == BSPGEMM_ERR_ALLOC);
        /* (the restatement's loader is NOT run on these: it restates the reference's, which trusts the file -- the
         * size-line loop of final/mmio.c:205-214 never ends on an unparsable token and final/coo2csc.c:37-56 indexes
         * with whatever the file says; rejecting such files is the product loader's job, which is what is tested) */
    }
    snprintf(path, sizeof path, "%s/does_not_exist.mtx", argv[1]);
    CHECK(bspgemm_readCOO(path, &lr, &lc, &M, &N, &nz) == BSPGEMM_ERR_IO);
    snprintf(path, sizeof path, "%s/no_such_dir/x.mtx", argv[1]);
    {
        const int r1[2] = {0, 1}, c1[1] = {0};
        CHECK(bspgemm_write_mtx(path, 1, 1, r1, c1) == BSPGEMM_ERR_IO);
    }
    /* prefault: unaligned start and length */
    {
        char *buf = malloc(3 * 4096 + 17);
        CHECK(buf);
        bspgemm_par_prefault(buf + 5, 3 * 4096 + 3);
        free(buf);
    }
    printf("host_asan ok: uniform nnz(C) = %lld\n", (long long)nnz_u);
    return 0;
}
