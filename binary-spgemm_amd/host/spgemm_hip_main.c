/*
 * spgemm_hip_main.c -- command-line driver, drop-in for the reference's benchmark binary
 *
 *     [mpirun -n P] SpGEMM_hip  path-to-matrix  threadslice_size  number_of_threads  times_to_run
 *
 * (reference: main + test_mpi, final/SpGEMM_mpi_omp.c:294-366; usage text :358).  Same four
 * positionals; `threadslice_size` (tBlock) and `number_of_threads` are accepted and echoed but
 * advisory -- the GPU grid replaces OpenMP slices and there is no divisibility rule
 * (README.md:14-17 of the reference).  Always A*A (:322).  Output on rank 0: the reference's CSV
 * line (:336)
 *     tasks,threads,tasks*threads,tBlock,path,n,nnz(A),nnz(C),mean,median,fastest
 * followed by extra columns: gnz_per_s,alg_GBps,roofline_frac,gpus,e2e_fastest.
 * The three times are seconds of the device-resident product (A already in HBM, result left in
 * HBM, C.row_ptr stitched), the region the reference times around SpGEMM_mpi (:320-324) minus its
 * host gathers; e2e_fastest adds the copy of C to host memory.  nnz(C) is printed as a 64-bit
 * integer (the reference's %d overflows above 2^31-1).
 *
 * Two optional positionals widen the reference's always-A*A driver:
 *     SpGEMM_hip  A.mtx  tBlock  threads  times  [B.mtx  [C_out.mtx]]
 * computes the FILE-orientation product A*B of two (possibly rectangular) pattern files and, when
 * a third path is given, writes C as `matrix coordinate pattern general`.  readCOO keeps every
 * file transposed in memory (final/utils.c:77), and (A*B)^T = B^T * A^T, so the in-memory
 * product is loaded(B) * loaded(A); the writer transposes back.  "-" as B.mtx means A.
 *
 * With -DBSPGEMM_WITH_MPI: one MPI rank per GPU (rank r -> device r mod #devices), every rank
 * reads the whole file (B replicated, like :309), rows are cut at equal work, C.row_ptr is
 * stitched by bspgemm_comm_stitch_row_ptr -- one all-gather of int32 row lengths over RCCL (unique
 * id broadcast with MPI_Bcast), or over MPI_Allgather when the ranks share a GPU (host/
 * mpi_transport.h); col_idx stays on the GPUs.
 * BSPGEMM_EXPAND_SYMMETRIC=1 in the environment loads symmetric files expanded (opt-in, f2).
 */
#include "../../include/bspgemm.h"

#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef BSPGEMM_WITH_MPI
#include "mpi_transport.h"
#endif

static double now_s(void)
{
    struct timespec t;
    clock_gettime(CLOCK_MONOTONIC, &t);                       /* tictoc, final/utils.c:104-113 */
    return (double)t.tv_sec + (double)t.tv_nsec * 1e-9;
}

static int cmp_double(const void *a, const void *b)
{
    const double x = *(const double *)a, y = *(const double *)b;
    return (x > y) - (x < y);
}

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
    /* before anything touches the HIP runtime: enough hardware queues for the library's three
     * streams next to RCCL's (tuning knobs) */
    setenv("GPU_MAX_HW_QUEUES", "8", 0);
    int numtasks = 1, rank = 0;
#ifdef BSPGEMM_WITH_MPI
    int provided;
    MPI_Init_thread(&argc, &argv, MPI_THREAD_FUNNELED, &provided);       /* :352 */
    MPI_Comm_size(MPI_COMM_WORLD, &numtasks);
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
#endif
    if (argc < 5 || argc > 7) {                                          /* :357-360 (+ optional B, C_out) */
        printf("usage: mpirun  -n  numtasks  SpGEMM_mpi_omp  path-to-matrix  threadslice_size  number_of_threads  times_to_run\n");
        exit(1);
    }
    const int tBlock = atoi(argv[2]);
    const int threads = atoi(argv[3]);
    int times = atoi(argv[4]);
    if (times < 1) times = 1;

    const char *sym = getenv("BSPGEMM_EXPAND_SYMMETRIC");
    const unsigned rflags = (sym && sym[0] == '1') ? BSPGEMM_READ_EXPAND_SYMMETRIC : 0u;
    uint32_t *Arow, *Acol, An, Am, Annz;
    bspgemm_status st = bspgemm_readCOO_ex(argv[1], rflags, &Arow, &Acol, &An, &Am, &Annz);   /* An = M, Am = N */
    /* the reference prints for a bad banner only (utils.c:56-59); fopen and size-line failures
     * exit(1) silently (:54-55, :60-61) */
    if (st == BSPGEMM_ERR_FORMAT) printf("Could not process Matrix Market banner.\n");
    if (st != BSPGEMM_OK) exit(1);

    const int ndev = bspgemm_device_count();
    const char *devenv = getenv("BSPGEMM_DEVICE");
    bspgemm_context *ctx;
    CHECK(bspgemm_create(devenv ? atoi(devenv) : (ndev > 0 ? rank % ndev : 0), &ctx), "bspgemm_create");

    /* in-memory (transposed) shapes: loaded(X) has X.N rows and X.M columns */
    const int a_rows = (int)Am, a_cols = (int)An;
    bspgemm_matrix *A, *L = NULL;                   /* L = left operand of the in-memory product */
    CHECK(bspgemm_matrix_upload(ctx, a_rows, a_cols, (const int *)Arow, (const int *)Acol, &A), "upload");
    int c_cols = a_cols;
    if (argc >= 6 && strcmp(argv[5], "-") != 0) {
        uint32_t *Brow, *Bcol, Bn, Bm, Bnnz;
        st = bspgemm_readCOO_ex(argv[5], rflags, &Brow, &Bcol, &Bn, &Bm, &Bnnz);
        if (st == BSPGEMM_ERR_FORMAT) printf("Could not process Matrix Market banner.\n");
        if (st != BSPGEMM_OK) exit(1);
        if ((int)Bn != a_rows) { fprintf(stderr, "inner dimensions differ: A is %ux%u, B is %ux%u\n", An, Am, Bn, Bm); exit(1); }
        CHECK(bspgemm_matrix_upload(ctx, (int)Bm, (int)Bn, (const int *)Brow, (const int *)Bcol, &L), "upload B");
        free(Brow); free(Bcol);
    }
    bspgemm_matrix *Lhs = L ? L : A;                 /* loaded(B) * loaded(A), or A*A like :322 */
    const int c_rows = bspgemm_matrix_rows(Lhs);

    int *bounds = malloc(((size_t)numtasks + 1) * sizeof(int));
    CHECK(bspgemm_partition_rows(ctx, Lhs, A, numtasks, bounds), "partition");
    const int r0 = bounds[rank], r1 = bounds[rank + 1];

#ifdef BSPGEMM_WITH_MPI
    bspgemm_comm *comm = NULL;
    const int64_t *d_row_ptr_global = NULL;   /* stitched C.row_ptr, device, owned by comm */
    int64_t *shard_nnz = malloc((size_t)numtasks * sizeof(int64_t));
    int used_rccl = 0;
    if (numtasks > 1) CHECK(mpi_make_comm(ctx, rank, numtasks, devenv ? 1 : ndev, &comm, &used_rccl), "communicator");
#endif

    double *alltimes = malloc((size_t)times * sizeof(double));
    double timesum = 0, e2e_fastest = 1e300;
    long long nnzC = 0;
    bspgemm_stats stats;
    memset(&stats, 0, sizeof stats);
    int64_t *hrow = malloc(((size_t)(r1 - r0) + 1) * sizeof(int64_t));
    (void)c_rows;
    for (int i = 0; i < times; i++) {
#ifdef BSPGEMM_WITH_MPI
        MPI_Barrier(MPI_COMM_WORLD);                                     /* :319 */
#endif
        const double t0 = now_s();                                       /* tic :320 */
        bspgemm_result *C;
        CHECK(bspgemm_multiply(ctx, Lhs, A, r0, r1, &C), "bspgemm_multiply");
        long long local_nnz = bspgemm_result_nnz(C);
        nnzC = local_nnz;
#ifdef BSPGEMM_WITH_MPI
        if (numtasks > 1) {
            CHECK(bspgemm_comm_stitch_row_ptr(comm, C, bounds, &d_row_ptr_global, shard_nnz), "stitch");
            nnzC = 0;
            for (int r = 0; r < numtasks; r++) nnzC += shard_nnz[r];
        }
#endif
        alltimes[i] = now_s() - t0;                                      /* toc :324 */
        timesum += alltimes[i];
        bspgemm_last_stats(ctx, &stats);
        /* end-to-end: bring this rank's shard of C to the host as well */
        int *hcol = malloc((size_t)(local_nnz > 0 ? local_nnz : 1) * sizeof(int));
        if (hcol) {
            CHECK(bspgemm_result_download(ctx, C, hrow, hcol), "download");
            const double e2e = now_s() - t0;
            if (e2e < e2e_fastest) e2e_fastest = e2e;
            if (argc == 7 && i == times - 1 && numtasks == 1)            /* f2: C in the file's orientation */
                CHECK(bspgemm_write_result_mtx(argv[6], c_rows, c_cols, hrow, hcol), "write C");
            free(hcol);                                                  /* isroot free(nCcol) :327 */
        }
        bspgemm_result_free(C);
    }

    const double mean = timesum / times;
    qsort(alltimes, (size_t)times, sizeof(double), cmp_double);         /* quickSortD :331 */
    const double median = alltimes[(times - 1) / 2];                    /* :332 */
    const double fastest = alltimes[0];
    long long bytes_alg = stats.bytes_alg;
#ifdef BSPGEMM_WITH_MPI
    if (numtasks > 1) {
        long long tot = 0;
        MPI_Reduce(&bytes_alg, &tot, 1, MPI_LONG_LONG, MPI_SUM, 0, MPI_COMM_WORLD);
        bytes_alg = tot;
    }
#endif
    if (rank == 0) {
        const double gnz = (double)nnzC / median / 1e9;
        const double gbps = (double)bytes_alg / median / 1e9;
        printf("%d,%d,%d,%d,%s,%u,%u,%lld,%lf,%lf,%lf,%.4f,%.1f,%.4f,%d,%lf\n", numtasks, threads,
               numtasks * threads, tBlock, argv[1], An, Annz, nnzC, mean, median, fastest,
               gnz, gbps, gbps / (8000.0 * numtasks), numtasks, e2e_fastest);
    }
    free(alltimes); free(hrow); free(bounds);
    free(Acol); free(Arow);                                              /* :339-340 */
    bspgemm_matrix_free(A);
    bspgemm_matrix_free(L);
#ifdef BSPGEMM_WITH_MPI
    if (comm) bspgemm_comm_destroy(comm);
    free(shard_nnz);
#endif
    bspgemm_destroy(ctx);
#ifdef BSPGEMM_WITH_MPI
    MPI_Finalize();                                                      /* :364 */
#endif
    return 0;
}
