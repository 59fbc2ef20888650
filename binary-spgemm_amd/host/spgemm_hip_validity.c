/*
 * spgemm_hip_validity.c -- drop-in for the reference's `make test` binary
 *
 *     [mpirun -n P] SpGEMM_hip_validity  path-to-matrix  threadslice_size  number_of_threads
 *
 * (reference: final/SpGEMM_mpi_omp_validity.c:308-375, run by final/Makefile:11-12 as
 *  `mpirun -n 4 SpGEMM_mpi_omp_validity ../Matlab/validity_test.mtx 6250 2`).
 * The reference multiplies once through the decomposed path (4 ranks x 2 threads x 6250-row
 * slices, :331) and once through the serial kernel over all rows (:337), compares the two CSRs
 * exactly (SpGEMM_valid, :290-302) and prints one of two messages (:340,:342); the exit status
 * is 0 either way.
 *
 * Plain build (SpGEMM_hip_validity, one process): the decomposed run is the product computed as
 * row shards of `threadslice_size` rows each (as many bspgemm_multiply calls as the reference has
 * slices, stitched on the host like :211-223); the whole run is the int32 drop-in SpGEMM_hip.
 *
 * -DBSPGEMM_WITH_MPI (SpGEMM_hip_validity_mpi, the mode of :308-353): every rank reads the file
 * (:323), the decomposed run is SpGEMM_hip_multi -- this library's SpGEMM_mpi: equal-work row
 * shards over the ranks, C.row_ptr stitched by one all-gather, col_idx gathered on rank 0 -- over
 * RCCL when every rank has a GPU, over MPI when the ranks share one (host/mpi_transport.h); rank 0
 * then runs the whole product through SpGEMM_hip (:334-337), compares and prints (:339-343).
 *
 * Both runs are GPU runs: this binary checks decomposition invariance like the reference's; the
 * CPU-vs-GPU parity lives in tests/.
 */
#include "../../include/bspgemm.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef BSPGEMM_WITH_MPI
#include "mpi_transport.h"
#endif

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
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    int numtasks = 1, rank = 0;
    (void)numtasks;
#ifdef BSPGEMM_WITH_MPI
    int provided;
    MPI_Init_thread(&argc, &argv, MPI_THREAD_FUNNELED, &provided);       /* :361 */
    MPI_Comm_size(MPI_COMM_WORLD, &numtasks);
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
#endif
    if (argc != 4) {                                                     /* :366-369 */
        printf("usage: mpirun  -n  numtasks  SpGEMM_mpi_omp  path-to-matrix  threadslice_size  number_of_threads\n");
        exit(1);
    }
    int tBlock = atoi(argv[2]);
    uint32_t *Arow, *Acol, An, Am, Annz;
    bspgemm_status st = bspgemm_readCOO(argv[1], &Arow, &Acol, &An, &Am, &Annz);
    if (st == BSPGEMM_ERR_FORMAT) printf("Could not process Matrix Market banner.\n");   /* utils.c:56-59; :54,:60 are silent */
    if (st != BSPGEMM_OK) exit(1);
    const int n = (int)An;
    if (tBlock <= 0 || tBlock > n) tBlock = n > 0 ? n : 1;

    const int ndev = bspgemm_device_count();
    const char *devenv = getenv("BSPGEMM_DEVICE");
    const int device = devenv ? atoi(devenv) : (ndev > 0 ? rank % ndev : 0);
    bspgemm_dropin_set_device(device);
    bspgemm_context *ctx;
    CHECK(bspgemm_create(device, &ctx), "bspgemm_create");

    int *nCrow = calloc((size_t)n + 1, sizeof(int)), *nCcol = NULL;      /* the decomposed result */
    int ok = 1;
#ifdef BSPGEMM_WITH_MPI
    bspgemm_comm *comm = NULL;
    int used_rccl = 0;
    CHECK(mpi_make_comm(ctx, rank, numtasks, devenv ? 1 : ndev, &comm, &used_rccl), "communicator");
    if (SpGEMM_hip_multi(comm, (int *)Acol, (int *)Arow, n, (int *)Acol, (int *)Arow, n, &nCcol, nCrow, tBlock) != 0)
        ok = 0;                                                          /* :331 */
    int all_ok = 0;
    MPI_Allreduce(&ok, &all_ok, 1, MPI_INT, MPI_MIN, MPI_COMM_WORLD);
    if (!all_ok) {
        if (rank == 0) printf("The results dont match\n");
        MPI_Finalize();
        return 0;
    }
#else
    {   /* one multiply per slice of tBlock rows, stitched on the host */
        bspgemm_matrix *A;
        CHECK(bspgemm_matrix_upload(ctx, n, n, (const int *)Arow, (const int *)Acol, &A), "upload");
        int64_t *srow = malloc(((size_t)tBlock + 1) * sizeof(int64_t));
        long long base = 0, cap = 1;
        nCcol = malloc(sizeof(int));                                     /* an empty product still has a Ccol */
        for (int r0 = 0; r0 < n && ok; r0 += tBlock) {
            const int r1 = r0 + tBlock < n ? r0 + tBlock : n;
            bspgemm_result *C;
            CHECK(bspgemm_multiply(ctx, A, A, r0, r1, &C), "bspgemm_multiply");
            const long long snnz = bspgemm_result_nnz(C);
            if (base + snnz > 0x7fffffffll) { ok = 0; bspgemm_result_free(C); break; }
            if (base + snnz > cap) {                                     /* grow like :28-31 */
                cap = (base + snnz) + (base + snnz) / 4 + 1024;
                int *grown = realloc(nCcol, (size_t)cap * sizeof(int));
                if (!grown) { ok = 0; bspgemm_result_free(C); break; }
                nCcol = grown;
            }
            CHECK(bspgemm_result_download(ctx, C, srow, nCcol + base), "download");
            for (int j = 1; j <= r1 - r0; j++) nCrow[r0 + j] = (int)(srow[j] + base);   /* rebase, :215-221 */
            base += snnz;
            bspgemm_result_free(C);
        }
        free(srow);
        bspgemm_matrix_free(A);
    }
#endif
    if (rank == 0) {
        /* whole product through the int32 drop-in (the role of the serial run, :334-337) */
        int *tCrow = calloc((size_t)n + 1, sizeof(int)), *tCcol = NULL;
        if (SpGEMM_hip((int *)Acol, (int *)Arow, n, (int *)Acol, (int *)Arow, n, &tCcol, tCrow, tBlock) != 0) ok = 0;
        if (ok && nCcol && tCcol && bspgemm_csr_equal(nCcol, nCrow, tCcol, tCrow, n))   /* :339-343 */
            printf("Results of serial and multricore are the same!\n");
        else
            printf("The results dont match\n");
        free(tCcol); free(tCrow);
    }
    free(nCcol); free(nCrow);
#ifdef BSPGEMM_WITH_MPI
    bspgemm_comm_destroy(comm);
#endif
    bspgemm_destroy(ctx);
    free(Acol); free(Arow);
#ifdef BSPGEMM_WITH_MPI
    MPI_Finalize();
#endif
    return 0;
}
