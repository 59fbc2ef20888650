/*
 * spgemm_hip_closure.c -- reflexive-transitive closure of a pattern matrix by repeated boolean
 * squaring, everything resident on the GPU (the application the reference's
 * report motivates the kernel with -- its old/BSpGEMM.c:75-126 keeps an OR-accumulating variant).
 *
 *     SpGEMM_hip_closure  A.mtx  [C_out.mtx]
 *
 * Prints: path,n,nnz(A),products_computed,nnz(closure),seconds.  A.mtx is read like every other
 * input (readCOO, final/utils.c:47-81: transposed in memory; the closure of the transpose is the
 * transpose of the closure, and the writer transposes back).
 */
#include "../../include/bspgemm.h"

#include <stdio.h>
#include <stdlib.h>
#include <time.h>

#define CHECK(st, what)                                                                         \
    do {                                                                                        \
        bspgemm_status s_ = (st);                                                               \
        if (s_ != BSPGEMM_OK) {                                                                 \
            fprintf(stderr, "%s: %s: %s\n", what, bspgemm_status_string(s_), bspgemm_last_error()); \
            exit(1);                                                                            \
        }                                                                                       \
    } while (0)

int main(int argc, char **argv)
{
    if (argc != 2 && argc != 3) {
        printf("usage: SpGEMM_hip_closure  path-to-matrix  [path-to-result]\n");
        exit(1);
    }
    uint32_t *Arow, *Acol, M, N, nnz;
    bspgemm_status st = bspgemm_readCOO(argv[1], &Arow, &Acol, &M, &N, &nnz);
    if (st == BSPGEMM_ERR_FORMAT) printf("Could not process Matrix Market banner.\n");
    if (st != BSPGEMM_OK) exit(1);
    if (M != N) { fprintf(stderr, "closure needs a square matrix (%ux%u)\n", M, N); exit(1); }
    const char *devenv = getenv("BSPGEMM_DEVICE");
    bspgemm_context *ctx;
    CHECK(bspgemm_create(devenv ? atoi(devenv) : 0, &ctx), "bspgemm_create");
    bspgemm_matrix *A;
    CHECK(bspgemm_matrix_upload(ctx, (int)M, (int)M, (const int *)Arow, (const int *)Acol, &A), "upload");
    struct timespec t0, t1;
    clock_gettime(CLOCK_MONOTONIC, &t0);
    bspgemm_result *T;
    int products = 0;
    CHECK(bspgemm_closure(ctx, A, 64, &T, &products), "bspgemm_closure");
    clock_gettime(CLOCK_MONOTONIC, &t1);
    const double secs = (double)(t1.tv_sec - t0.tv_sec) + (double)(t1.tv_nsec - t0.tv_nsec) * 1e-9;
    printf("%s,%u,%u,%d,%lld,%lf\n", argv[1], M, nnz, products, (long long)bspgemm_result_nnz(T), secs);
    if (argc == 3) {
        const long long tn = bspgemm_result_nnz(T);
        int64_t *rp = malloc(((size_t)M + 1) * sizeof(int64_t));
        int *ci = malloc((size_t)(tn > 0 ? tn : 1) * sizeof(int));
        if (!rp || !ci) exit(1);
        CHECK(bspgemm_result_download(ctx, T, rp, ci), "download");
        CHECK(bspgemm_write_result_mtx(argv[2], (int)M, (int)M, rp, ci), "write");
        free(rp); free(ci);
    }
    bspgemm_result_free(T);
    bspgemm_matrix_free(A);
    bspgemm_destroy(ctx);
    free(Arow); free(Acol);
    return 0;
}
