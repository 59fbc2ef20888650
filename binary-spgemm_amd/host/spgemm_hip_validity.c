/*
 * spgemm_hip_validity.c -- drop-in for the reference's `make test` binary
 *
 *     SpGEMM_hip_validity  path-to-matrix  threadslice_size  number_of_threads
 *
 * (reference: final/SpGEMM_mpi_omp_validity.c:308-375, run by final/Makefile:11-12 as
 *  `mpirun -n 4 SpGEMM_mpi_omp_validity ../Matlab/validity_test.mtx 6250 2`).
 * The reference multiplies once through the decomposed path (4 ranks x 2 threads x 6250-row
 * slices, :331) and once through the serial kernel over all rows (:337), compares the two CSRs
 * exactly (SpGEMM_valid, :290-302) and prints one of two messages (:340,:342); the exit status
 * is 0 either way.  Here the decomposed run is the product computed as row shards of
 * `threadslice_size` rows each (as many bspgemm_multiply calls as the reference has slices,
 * stitched on the host like :211-223), the whole run is the int32 drop-in SpGEMM_hip over all
 * rows, and the same comparator and messages follow.  Both runs are GPU runs: this binary checks
 * decomposition invariance like the reference's; the CPU-vs-GPU parity lives in tests/.
 */
#include "../../include/bspgemm.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>

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
    if (argc != 4) {                                                     /* :366-369 */
        printf("usage: mpirun  -n  numtasks  SpGEMM_mpi_omp  path-to-matrix  threadslice_size  number_of_threads\n");
        exit(1);
    }
    int tBlock = atoi(argv[2]);
    uint32_t *Arow, *Acol, An, Am, Annz;
    bspgemm_status st = bspgemm_readCOO(argv[1], &Arow, &Acol, &An, &Am, &Annz);
    if (st == BSPGEMM_ERR_FORMAT) printf("Could not process Matrix Market banner.\n");
    if (st != BSPGEMM_OK) exit(1);
    const int n = (int)An;
    if (tBlock <= 0 || tBlock > n) tBlock = n > 0 ? n : 1;

    /* whole product through the int32 drop-in (the role of the serial run, :334-337) */
    int *tCrow = calloc((size_t)n + 1, sizeof(int)), *tCcol = NULL;
    if (SpGEMM_hip((int *)Acol, (int *)Arow, n, (int *)Acol, (int *)Arow, n, &tCcol, tCrow, tBlock) != 0) exit(1);

    /* decomposed product: one multiply per slice of tBlock rows, stitched on the host */
    const char *devenv = getenv("BSPGEMM_DEVICE");
    bspgemm_context *ctx;
    CHECK(bspgemm_create(devenv ? atoi(devenv) : 0, &ctx), "bspgemm_create");
    bspgemm_matrix *A;
    CHECK(bspgemm_matrix_upload(ctx, n, n, (const int *)Arow, (const int *)Acol, &A), "upload");
    int *nCrow = calloc((size_t)n + 1, sizeof(int));
    int *nCcol = malloc((size_t)(tCrow[n] > 0 ? tCrow[n] : 1) * sizeof(int));
    int64_t *srow = malloc(((size_t)tBlock + 1) * sizeof(int64_t));
    long long base = 0;
    int ok = 1;
    for (int r0 = 0; r0 < n && ok; r0 += tBlock) {
        const int r1 = r0 + tBlock < n ? r0 + tBlock : n;
        bspgemm_result *C;
        CHECK(bspgemm_multiply(ctx, A, A, r0, r1, &C), "bspgemm_multiply");
        const long long snnz = bspgemm_result_nnz(C);
        if (base + snnz > tCrow[n]) { ok = 0; bspgemm_result_free(C); break; }
        CHECK(bspgemm_result_download(ctx, C, srow, nCcol + base), "download");
        for (int j = 1; j <= r1 - r0; j++) nCrow[r0 + j] = (int)(srow[j] + base);   /* rebase, :215-221 */
        base += snnz;
        bspgemm_result_free(C);
    }
    if (ok && bspgemm_csr_equal(nCcol, nCrow, tCcol, tCrow, n))          /* :339-343 */
        printf("Results of serial and multricore are the same!\n");
    else
        printf("The results dont match\n");

    free(srow); free(nCcol); free(nCrow); free(tCcol); free(tCrow);
    bspgemm_matrix_free(A);
    bspgemm_destroy(ctx);
    free(Acol); free(Arow);
    return 0;
}
