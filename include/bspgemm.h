/*
 * bspgemm.h -- C ABI of libbspgemm.so: boolean (pattern-only) SpGEMM  C = A*B  on MI355X.
 *
 * This is the drop-in boundary for the reference's row-wise Gustavson path
 * (pavlidic/Binary-SpGEMM, final/SpGEMM_mpi_omp.c).  The reference has no plugin or FFI
 * registry: its boundary is a handful of plain C functions on raw `int*` CSR arrays plus the
 * command line.  Every entry point below names the reference interface it
 * replaces (file:line, relative to the reference checkout).  Plain pointers and sizes only; no
 * C++/torch types.  All functions are callable from C (the host side of this project is C).
 *
 * Conventions shared with the reference:
 *   - CSR, 0-based; `row_ptr[rows+1]` ascending; `col_idx[nnz]`; pattern only (no values).
 *   - argument order "col before row" in the drop-in signatures (final/SpGEMM_mpi_omp.c:15-18).
 *   - inputs need not have sorted or duplicate-free rows; outputs always have strictly
 *     ascending col_idx per row (the reference sorts every row, :47).
 * Differences: the native API returns C.row_ptr as int64 (the reference's
 * `int` overflows above 2^31-1 output nonzeros); the int32 drop-ins REFUSE (status
 * BSPGEMM_ERR_OVERFLOW, nothing written) instead of wrapping.
 */
#ifndef BSPGEMM_H
#define BSPGEMM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---------------------------------------------------------------- status ------------- */
typedef enum bspgemm_status {
    BSPGEMM_OK            = 0,
    BSPGEMM_ERR_INVALID   = 1,   /* bad argument (null pointer, negative size, row range)        */
    BSPGEMM_ERR_ALLOC     = 2,   /* host or device allocation failed                            */
    BSPGEMM_ERR_HIP       = 3,   /* a HIP runtime call failed (see bspgemm_last_error)          */
    BSPGEMM_ERR_NO_DEVICE = 4,   /* no usable gfx950 device: the product has NO CPU fallback    */
    BSPGEMM_ERR_OVERFLOW  = 5,   /* result does not fit the int32 drop-in interface             */
    BSPGEMM_ERR_IO        = 6,   /* file open / parse failure                                   */
    BSPGEMM_ERR_FORMAT    = 7,   /* Matrix Market banner rejected                               */
    BSPGEMM_ERR_COMM      = 8,   /* RCCL failure in the multi-GPU exchange                      */
    BSPGEMM_ERR_SIZE      = 9    /* Matrix Market size line or an entry rejected (the reference
                                    exits without a message there, final/utils.c:60-61)         */
} bspgemm_status;

const char *bspgemm_status_string(bspgemm_status s);
/* text of the last failure on this thread (HIP error string, file:line) */
const char *bspgemm_last_error(void);
/* what this build of the library contains (one line of text).  The shipped library has no
 * timing-only ablation paths: tests/test_abi.py asserts "BSP_ABLATE=0" here.                 */
const char *bspgemm_build_info(void);

/* ---------------------------------------------------------------- native handle API ---
 * Replaces, with device-resident operands and int64 row_ptr, the call
 *     SpGEMM_mpi(Acol,Arow,An, Acol,Arow,An, &nCcol,nCrow,tBlock)   final/SpGEMM_mpi_omp.c:322
 * and the functions under it: SpGEMM_omp :71-143, SpGEMM_bigslice :15-58 (+ quickSort,
 * final/utils.c:159-173, which has no GPU counterpart: rows are emitted in order).
 * Upload once, multiply `times` times, download if wanted -- the reference's timed region
 * (:320-324) likewise excludes I/O and CSR construction.                                     */
/* Threads: a bspgemm_context is NOT thread-safe -- it owns one set of workspaces, one pinned block
 * of read-back scalars and the timing slots of its last 16 multiplies; calls on one context must
 * be serialised by the caller (different contexts, also on one device, are independent).  The
 * int32 drop-ins share one process-wide context behind a mutex.
 * Memory: results freed with bspgemm_result_free go to a per-context cache (at most 8 buffers and a
 * quarter of the device memory) and are handed to the next multiply instead of hipMalloc; the
 * default flow additionally keeps a workspace of F entries (F = products), i.e. about 2F ints live
 * per context after a multiply (10.7 GB for BASELINE config 3).  bspgemm_destroy releases all.
 * Environment (read once, in bspgemm_create; every knob also has a setter, bspgemm_set_option / _set_flow /
 * _set_class_timing, which is what a running program uses): BSPGEMM_FLOW=auto|upper-bound|exact,
 * BSPGEMM_CLASS_STREAMS=1..3, BSPGEMM_CLASS_TIMING=0|1, BSPGEMM_RW_BLK=0|1, BSPGEMM_CHECK, BSPGEMM_SMALL=0|1, BSPGEMM_PAD_ROWS=0|1,
 * BSPGEMM_DEBUG_ALLOC, BSPGEMM_DROPIN_TIMING; BSPGEMM_DEVICE picks the drop-ins' device.  (BSPGEMM_RANK_ROWS=0|1|2 is a
 * development switch of the rank class, read once per process: 0 none, 1 default, 2 also for single-window column counts.)        */
typedef struct bspgemm_context bspgemm_context;   /* one per GPU: device, stream, workspaces  */
typedef struct bspgemm_matrix  bspgemm_matrix;    /* device-resident CSR operand, int32 row_ptr */
typedef struct bspgemm_result  bspgemm_result;    /* device-resident CSR product, int64 row_ptr */

int            bspgemm_device_count(void);       /* visible HIP devices (0 without a GPU)      */
bspgemm_status bspgemm_create(int device, bspgemm_context **ctx);
void           bspgemm_destroy(bspgemm_context *ctx);
/* run on an existing HIP stream (hipStream_t passed as void*; NULL = the context's own) */
bspgemm_status bspgemm_set_stream(bspgemm_context *ctx, void *hip_stream);
bspgemm_status bspgemm_synchronize(bspgemm_context *ctx);

/* Copy a host CSR to the device.  `row_ptr` may be an interior pointer into a larger matrix
 * (values absolute into `col_idx`, like &Arow[rank*tasksize] at final/SpGEMM_mpi_omp.c:171);
 * only col_idx[row_ptr[0] .. row_ptr[rows]) is transferred and the pointers are rebased.     */
bspgemm_status bspgemm_matrix_upload(bspgemm_context *ctx, int rows, int cols,
                                     const int *row_ptr, const int *col_idx,
                                     bspgemm_matrix **out);
/* Adopt device arrays the caller owns (not freed by bspgemm_matrix_free), row_ptr[0] == 0.
 * An operand is IMMUTABLE while the handle lives: the library keeps tables derived from row_ptr and col_idx
 * (row lengths by the byte, blocked extents, a copy of col_idx with every row on a 64-byte boundary) and builds them
 * on first use.  A caller that rewrites
 * the wrapped arrays in place must call bspgemm_matrix_invalidate before the next multiply; a
 * stale table would size rows from old lengths.                                                */
bspgemm_status bspgemm_matrix_wrap_device(bspgemm_context *ctx, int rows, int cols, int64_t nnz,
                                          const int *d_row_ptr, const int *d_col_idx,
                                          bspgemm_matrix **out);
/* drops the derived tables of an operand (they are rebuilt on the next use); synchronises the
 * context's stream first                                                                        */
bspgemm_status bspgemm_matrix_invalidate(bspgemm_matrix *m);
void    bspgemm_matrix_free(bspgemm_matrix *m);
int     bspgemm_matrix_rows(const bspgemm_matrix *m);
int     bspgemm_matrix_cols(const bspgemm_matrix *m);
int64_t bspgemm_matrix_nnz(const bspgemm_matrix *m);

/* Rows [row_begin,row_end) of C = A*B, the job of SpGEMM_bigslice(..., start_row, end_row)
 * (final/SpGEMM_mpi_omp.c:15-58): result row_ptr is slice-local (row_ptr[0] = 0, :26).
 * B.rows must be >= A.cols.  Asynchronous up to the internal size read-backs; the result is
 * complete on the context's stream when the call returns.                                    */
bspgemm_status bspgemm_multiply(bspgemm_context *ctx,
                                const bspgemm_matrix *A, const bspgemm_matrix *B,
                                int row_begin, int row_end, bspgemm_result **C);

/* How bspgemm_multiply gets from row sizes to C.col_idx.  Both give the same CSR, bit for bit.
 *   UPPER_BOUND  rows are placed by their product count F_i (an upper bound of |C_i|) in a workspace
 *                of F entries and squeezed into C.col_idx once the counts are scanned -- the GPU
 *                form of the reference's append-then-concatenate (final/SpGEMM_mpi_omp.c:38-42,
 *                110-131); holds 2F entries.
 *   EXACT        a symbolic pass sizes every row exactly first (the accumulate kernels without their emit half), C.row_ptr
 *                is their scan, and the numeric pass emits every row at its final place: nnz(C)
 *                entries, no F-sized workspace.
 *   AUTO         (default, or env BSPGEMM_FLOW=auto|upper-bound|exact) UPPER_BOUND, and EXACT when its
 *                buffers cannot be allocated.
 * (Round 3 also shipped a third, row-ordered single-pass flow, "FUSED"; it was 4x slower on large products
 * and was taken out in round 4 -- DESIGN.md section 2.1 keeps its measurements.)                  */
#define BSPGEMM_FLOW_AUTO        0
#define BSPGEMM_FLOW_UPPER_BOUND 1
#define BSPGEMM_FLOW_EXACT       2
bspgemm_status bspgemm_set_flow(bspgemm_context *ctx, int flow);

/* Per-class launch brackets (bspgemm_stats: ms_bin, ms_bin_count, t_bin, t_bin_count).  OFF by default (or env
 * BSPGEMM_CLASS_TIMING=1): an event pair around each of a multiply's ~17 class launches keeps consecutive
 * launches of a stream apart (measured: +0.06 ms on BASELINE config 3).  The phase times (ms_prepass ..
 * ms_stitch) are always recorded.  A profiling pass switches this on for the multiplies it wants itemised. */
bspgemm_status bspgemm_set_class_timing(bspgemm_context *ctx, int on);

/* The remaining tuning knobs as calls (the environment variables of the same names are only their initial
 * values).  None of them changes a result, only which kernels produce it -- and bspgemm_stats says which did
 * (prepass_kernel, class_streams, flow, small_path), so that a test can tell that the path it asked for ran.
 *   CLASS_STREAMS    1..3  HIP streams the capacity-class launches of a phase alternate over (default 2)
 *   BLOCKED_EXTENTS  -1 decide per operand (default: B of 2^21 rows or more and not dominated by rows of
 *                    255+ nonzeros), 0 never, 1 always: whether the prepass gathers B's blocked extents table
 *                    instead of B.row_ptr pairs.  Decided when an operand is first used as B, so set it
 *                    before that (or call bspgemm_matrix_invalidate on the operand).
 *   CHECK            0/1   debug checks: the exact flow never emits on unverified sizes, and the accumulate
 *                    kernels verify each row's gathered product count against its capacity class (a stale
 *                    derived table -- see bspgemm_matrix_invalidate -- then fails the multiply with
 *                    BSPGEMM_ERR_INVALID instead of overrunning LDS)
 *   PADDED_ROWS      0 never (default), 1 always, -1 decide per operand (B of 2^20 nonzeros or more with a mean row length
 *                    of 8 or more): the accumulate kernels gather B's rows from a derived copy of B.col_idx in which every
 *                    row starts on a 64-byte boundary (a row then costs ceil(len / 16) 64-byte sectors instead of one
 *                    more; +1.4 x nnz(B) ints of device memory on the bench matrix).  Opt-in: measured worth 8-9 % of the
 *                    numeric phase on matrices whose rows all have 16 entries and nothing on the R-MAT bench matrix
 *                    (DESIGN.md 4.2).  Decided on first use as B, like BLOCKED_EXTENTS.
 *   SMALL_PATH       -1 automatic (default: products of at most 65536 with a cached result buffer take the
 *                    single-launch path), 0 never, 1 whenever the product fits it                       */
typedef enum bspgemm_option {
    BSPGEMM_OPT_CLASS_STREAMS   = 1,
    BSPGEMM_OPT_BLOCKED_EXTENTS = 2,
    BSPGEMM_OPT_CHECK           = 3,
    BSPGEMM_OPT_SMALL_PATH      = 4,
    BSPGEMM_OPT_PADDED_ROWS     = 5
} bspgemm_option;
bspgemm_status bspgemm_set_option(bspgemm_context *ctx, bspgemm_option opt, int value);
/* current value of a knob (INT32_MIN for an unknown option or a NULL context) */
int            bspgemm_get_option(const bspgemm_context *ctx, bspgemm_option opt);
/* 1 if products with `m` as B gather its blocked extents table, 0 if they gather B.row_ptr pairs, -1 if
 * that has not been decided yet (the operand has not been used as B since it was created / invalidated) */
int            bspgemm_matrix_uses_blocked_table(const bspgemm_matrix *m);
/* the same for the padded copy of col_idx (BSPGEMM_OPT_PADDED_ROWS) */
int            bspgemm_matrix_uses_padded_rows(const bspgemm_matrix *m);

/* C = F .* (A*B), complement convention of SpGEMM_masked (final/SpGEMM_mpi_omp.c:232-288):
 * a column k is admitted to row i only if (i,k) is in F's pattern.                           */
bspgemm_status bspgemm_multiply_masked(bspgemm_context *ctx,
                                       const bspgemm_matrix *A, const bspgemm_matrix *B,
                                       const bspgemm_matrix *F,
                                       int row_begin, int row_end, bspgemm_result **C);

int            bspgemm_result_rows(const bspgemm_result *C);
int64_t        bspgemm_result_nnz(const bspgemm_result *C);
const int64_t *bspgemm_result_row_ptr_device(const bspgemm_result *C);   /* rows+1 entries  */
const int     *bspgemm_result_col_idx_device(const bspgemm_result *C);   /* nnz entries     */
/* copy to host; either pointer may be NULL to skip that array                                */
bspgemm_status bspgemm_result_download(bspgemm_context *ctx, const bspgemm_result *C,
                                       int64_t *row_ptr, int *col_idx);
void           bspgemm_result_free(bspgemm_result *C);

/* A product becomes the next operand without leaving the GPU (int32 row_ptr copy; fails with
 * BSPGEMM_ERR_OVERFLOW above 2^31-1 nonzeros).  `cols` = number of columns of C (= B.cols).      */
bspgemm_status bspgemm_matrix_from_result(bspgemm_context *ctx, const bspgemm_result *C, int cols,
                                          bspgemm_matrix **out);

/* Reflexive-transitive closure by repeated boolean squaring, everything device-resident -- the
 * application the reference's report motivates the kernel with (its old/BSpGEMM.c:75-126 keeps
 * an OR-accumulating variant for it): T0 = A or I, T(k+1) = T(k)*T(k) until nnz stops growing
 * (at most max_iter products; ceil(log2 n) suffice).  A must be square.  *iterations = products
 * computed.  The result is T as a product object.                                              */
bspgemm_status bspgemm_closure(bspgemm_context *ctx, const bspgemm_matrix *A, int max_iter,
                               bspgemm_result **T, int *iterations);

/* Per-row work F_i = sum_{j in A_i} |B_j| ("products", the flag probes of :36-38) as an
 * exclusive prefix over rows [0,A.rows]: prefix[rows] = F.  Used to cut GPU shards at equal
 * work instead of equal row counts (the reference cuts An/numtasks rows, :165). */
bspgemm_status bspgemm_row_work_prefix(bspgemm_context *ctx,
                                       const bspgemm_matrix *A, const bspgemm_matrix *B,
                                       int64_t *prefix_host /* A.rows+1 */);
/* bounds[0..parts] with bounds[0]=0, bounds[parts]=A.rows, equal-F contiguous shards         */
bspgemm_status bspgemm_partition_rows(bspgemm_context *ctx,
                                      const bspgemm_matrix *A, const bspgemm_matrix *B,
                                      int parts, int *bounds);

/* counters of the last bspgemm_multiply* on this context */
#define BSPGEMM_MAX_BINS 20
typedef struct bspgemm_stats {
    int64_t rows;            /* rows multiplied                                              */
    int64_t nnz_a;           /* A nonzeros in those rows                                     */
    int64_t products;        /* F                                                            */
    int64_t nnz_c;           /* output nonzeros                                              */
    int64_t bytes_alg;       /* 4(rows+1)+4nnzA+8nnzA+4F+4nnzC+8(rows+1)     */
    int64_t bytes_read_alg;  /* its HBM-read part: bytes_alg - 4nnzC - 8(rows+1)             */
    int64_t rows_per_bin[BSPGEMM_MAX_BINS]; /* rows per capacity class: [0] empty rows,
                                [1..bins-4] one-wavefront rows with at most bin_cap[b] products,
                                [bins-3] rank rows (2048 < products <= bin_cap: one 512-thread
                                workgroup with a rank bitmap; bin_cap = 2048 where the class is
                                not used), [bins-2], [bins-1] heavy rows (one 512- / 1024-thread
                                workgroup each over dense column windows); rest unused        */
    float   ms_total;        /* hipEvent time of the whole multiply on the stream            */
    float   ms_symbolic;     /* = ms_prepass + ms_count: everything that sizes C.row_ptr     */
    float   ms_prepass;      /*   row work (products per row) + scan + capacity classes      */
    float   ms_count;        /*   exact row sizes (count pass of the one-wave classes, heavy rows) + scan */
    float   ms_numeric;      /* accumulate + emit kernels, rows written at their final place */
    float   ms_stitch;       /* what is left exposed after them (heavy-row move; masked
                                product: count scan + compaction)                            */
    float   ms_bin[BSPGEMM_MAX_BINS];       /* per class: its numeric-phase launch (0 unless bspgemm_set_class_timing) */
    float   ms_bin_count[BSPGEMM_MAX_BINS]; /* per class: its symbolic-phase (count) launch  */
    float   t_bin[BSPGEMM_MAX_BINS];        /* ... and when those launches STARTED, in ms    */
    float   t_bin_count[BSPGEMM_MAX_BINS];  /*     since the multiply began (the class launches
                                               alternate over two streams: with the durations
                                               above this is their timeline)                 */
    int     bins;            /* classes in use, including [0] and the heavy class            */
    int     bin_cap[BSPGEMM_MAX_BINS];      /* products a row of class b may have (masked product:
                                mask-row length); 0 for [0], INT32_MAX for the heavy class   */
    /* which path produced the result (so that a test of a knob can assert that the knob took) */
    int     flow;            /* BSPGEMM_FLOW_UPPER_BOUND or BSPGEMM_FLOW_EXACT: the flow that ran     */
    int     prepass_kernel;  /* 0 k_row_work (B.row_ptr pairs), 1 k_row_work_blk (blocked extents table),
                                2 the single-launch small path's own prepass                          */
    int     class_streams;   /* streams the class launches alternated over                           */
    int     small_path;      /* 1: the single-launch path for small products ran                      */
    int     checked;         /* 1: the device-side capacity guard was armed (BSPGEMM_OPT_CHECK)       */
    int     padded_rows;     /* 1: B's rows were gathered from the padded copy of B.col_idx           */
} bspgemm_stats;
bspgemm_status bspgemm_last_stats(const bspgemm_context *ctx, bspgemm_stats *out);
/* ... and of earlier ones: age 0 = the last multiply, 1 = the one before, ... up to 15.  The
 * HIP events of a multiply are its own, so K timed steps can be read back after the timed region. */
bspgemm_status bspgemm_stats_at(const bspgemm_context *ctx, int age, bspgemm_stats *out);

/* ---------------------------------------------------------------- int32 drop-ins ------
 * Same argument lists and ownership as the reference functions they replace: inputs are host
 * arrays and are not modified; `Crow` is caller memory with An+1 ints; `*Ccol` is allocated
 * here with libc malloc so the caller's free() (final/SpGEMM_mpi_omp.c:327) stays valid.
 * The reference functions return void and check nothing; these return a status as well and,
 * on failure, leave *Ccol = NULL and print one line to stderr -- they never fall back to a CPU
 * path.  `tBlock` is accepted and ignored (the GPU grid replaces OpenMP slices).             */

/* replaces SpGEMM_omp, final/SpGEMM_mpi_omp.c:71-74 (all An rows; no divisibility rule) */
int SpGEMM_hip(int *Acol, int *Arow, int An,
               int *Bcol, int *Brow, int Bm,
               int **Ccol, int *Crow, int tBlock);

/* replaces SpGEMM_bigslice, final/SpGEMM_mpi_omp.c:15-18: rows [start_row,end_row), slice-local
 * Crow, *Ccol pre-allocated by the caller with *Csize ints and grown with realloc if needed   */
int SpGEMM_hip_bigslice(int *Acol, int *Arow, int An,
                        int *Bcol, int *Brow, int Bm,
                        int **Ccol, int *Crow, int *Csize,
                        int start_row, int end_row);

/* replaces SpGEMM_mat, Matlab/inc/BSpGEMM.h:2-4 (coder.ceval target, Matlab/SpGEMM.m:9-11):
 * Ccol pre-allocated by the caller with the true nnz                                          */
int SpGEMM_hip_mat(int *Acol, int *Arow, int An,
                   int *Bcol, int *Brow, int Bm,
                   int *Ccol, int *Crow);

/* replaces SpGEMM_masked, final/SpGEMM_mpi_omp.c:232-235 */
int SpGEMM_hip_masked(int *Acol, int *Arow, int An,
                      int *Bcol, int *Brow, int Bm,
                      int *Fcol, int *Frow,
                      int **Ccol, int *Crow, int *Csize);

/* device used by the drop-ins (default 0, or env BSPGEMM_DEVICE) */
int bspgemm_dropin_set_device(int device);

/* ---------------------------------------------------------------- multi-GPU -----------
 * Replaces SpGEMM_mpi, final/SpGEMM_mpi_omp.c:155-225: one process (or thread) per GPU, B
 * replicated, contiguous A-row shards.  The reference gathers nnz, Ccol and Crow on rank 0
 * with MPI_Reduce/Gather/Gatherv (:178-204) and rebases Crow serially (:211-223); here every
 * GPU all-gathers the shard sizes and its local row_ptr over RCCL and rebases on device, so
 * every rank ends with the global C.row_ptr; col_idx stays sharded on the GPUs.
 * The communicator is built from an RCCL unique id that the launcher distributes (MPI,
 * torch.distributed, a file ...): bspgemm_comm_unique_id on rank 0, then bspgemm_comm_create
 * on every rank with the same 128 bytes.                                                      */
typedef struct bspgemm_comm bspgemm_comm;
#define BSPGEMM_UNIQUE_ID_BYTES 128
bspgemm_status bspgemm_comm_unique_id(unsigned char id[BSPGEMM_UNIQUE_ID_BYTES]);
bspgemm_status bspgemm_comm_create(bspgemm_context *ctx, const unsigned char id[BSPGEMM_UNIQUE_ID_BYTES],
                                   int rank, int nranks, bspgemm_comm **comm);
/* The same stitch over a transport the HOST supplies instead of RCCL -- for ranks that share one
 * GPU (RCCL refuses two ranks on a device; `mpirun -n 4` on a one-GPU box is how the reference's
 * `make test` runs, final/Makefile:11-12) and for tests.  Buffers are host memory.  Both
 * callbacks return 0 on success.  allgather: every rank contributes `bytes` bytes, recv gets
 * nranks*bytes, rank-major (MPI_Allgather).  gatherv: rank r contributes send_bytes ==
 * recv_bytes[r]; on `root`, recv gets the concatenation in rank order (MPI_Gatherv, what
 * final/SpGEMM_mpi_omp.c:203 does with Ccol); may be NULL when bspgemm_comm_gather_col_idx /
 * SpGEMM_hip_multi are not used.                                                               */
typedef struct bspgemm_host_transport {
    void *user;
    int (*allgather)(void *user, const void *send, void *recv, size_t bytes);
    int (*gatherv)(void *user, const void *send, size_t send_bytes, void *recv, const size_t *recv_bytes, int root);
} bspgemm_host_transport;
bspgemm_status bspgemm_comm_create_host(bspgemm_context *ctx, const bspgemm_host_transport *transport,
                                        int rank, int nranks, bspgemm_comm **comm);
void           bspgemm_comm_destroy(bspgemm_comm *comm);
/* Collective: every rank passes its own status, every rank gets the worst one.  Run it before a
 * collective that a failed rank would skip -- a rank that leaves the protocol alone leaves the others
 * blocked (the reference's MPI_Gather/Gatherv, final/SpGEMM_mpi_omp.c:178-204, have no such guard).
 * Every RCCL wait of this library is bounded (env BSPGEMM_COMM_TIMEOUT_S, default 120): on a timeout
 * or an asynchronous RCCL error the communicator is aborted and the call returns BSPGEMM_ERR_COMM.  */
bspgemm_status bspgemm_comm_agree(bspgemm_comm *comm, bspgemm_status mine);
/* After such a failure the communicator is DEAD: every later collective call on it (agree, stitch, gather,
 * SpGEMM_hip_multi) returns BSPGEMM_ERR_COMM at once, without touching the transport; destroy it and build a new one.
 * test hook (one-shot): 1 = the next SpGEMM_hip_multi on rank 0 behaves as if its host allocation had failed,
 * 2 = the next bounded RCCL wait behaves as if it had run out (the communicator is aborted), 3 = the next growth of the
 * stitch's staging buffers fails, 4 = the root's device buffer of the next RCCL col_idx gather cannot be allocated */
void           bspgemm_comm_inject_failure(bspgemm_comm *comm, int what);
int            bspgemm_comm_rank(const bspgemm_comm *comm);
int            bspgemm_comm_size(const bspgemm_comm *comm);
/* Second half of a stitch whose collective ran elsewhere (bspgemm/dist.py: torch.distributed):
 * `d_lengths` holds the all-gathered int32 row lengths, rank-major, `width` slots per rank of
 * which bounds[r+1]-bounds[r] are used; writes the global int64 row_ptr (bounds[nranks]+1
 * entries) on the device.  Enqueued on `hip_stream` (a hipStream_t; NULL = HIP's default
 * stream) -- pass the stream the collective was issued on; the context's own stream is not used.
 * Replaces the serial rebase of final/SpGEMM_mpi_omp.c:213-223.                               */
bspgemm_status bspgemm_lengths_to_row_ptr(bspgemm_context *ctx, const int *d_lengths, int nranks, int width,
                                          const int *bounds, int64_t *d_row_ptr, void *hip_stream);
/* The whole stitch: `bounds[nranks+1]` are the shard row bounds every rank used; `local` is this
 * rank's product of rows [bounds[rank],bounds[rank+1]).  Every rank contributes its rows' int32
 * lengths padded to the longest shard, ONE all-gather (RCCL, or the host transport), and
 * bspgemm_lengths_to_row_ptr on the gathered lengths.  On return *d_row_ptr_global points at a
 * device buffer owned by `comm` (bounds[nranks]+1 int64, valid until the next stitch or
 * bspgemm_comm_destroy) holding the global row_ptr on every rank, and shard_nnz[nranks] (host,
 * may be NULL) the per-shard nnz.  Replaces MPI_Reduce + MPI_Gather + MPI_Gather + the serial
 * rebase of final/SpGEMM_mpi_omp.c:178-223 (col_idx stays sharded on the GPUs).               */
bspgemm_status bspgemm_comm_stitch_row_ptr(bspgemm_comm *comm, const bspgemm_result *local,
                                           const int *bounds, const int64_t **d_row_ptr_global,
                                           int64_t *shard_nnz);
/* Root gather of the sharded col_idx, the MPI_Gatherv of final/SpGEMM_mpi_omp.c:203: on `root`,
 * col_idx_host[sum(shard_nnz)] receives the shards in rank order; other ranks pass NULL.
 * shard_nnz as returned by bspgemm_comm_stitch_row_ptr.  Collective: every rank calls it.      */
bspgemm_status bspgemm_comm_gather_col_idx(bspgemm_comm *comm, const bspgemm_result *local,
                                           const int64_t *shard_nnz, int root, int *col_idx_host);
/* replaces SpGEMM_mpi, final/SpGEMM_mpi_omp.c:155-158: the reference's argument list behind the
 * communicator it takes implicitly (MPI_COMM_WORLD).  Every rank passes the whole A and B; the
 * result (*Ccol malloc'ed here, Crow[An+1] caller memory) is valid on rank 0 only (:200-223);
 * other ranks get *Ccol = NULL.  Collective.  Returns a status like the other drop-ins.        */
int SpGEMM_hip_multi(bspgemm_comm *comm, int *Acol, int *Arow, int An,
                     int *Bcol, int *Brow, int Bm,
                     int **Ccol, int *Crow, int tBlock);

/* ---------------------------------------------------------------- host utilities (C) --
 * Plain C, no GPU needed.                                                                    */

/* replaces readCOO, final/utils.c:47-81 (with mm_read_banner final/mmio.c:96-179,
 * mm_read_mtx_crd_size :189-217 and coo2csc final/coo2csc.c:22-64): Matrix Market pattern
 * file -> CSR of the TRANSPOSED file matrix (the reference's argument swap at utils.c:77),
 * entries kept in file order inside each row, duplicates kept, square assumed (n = M).
 * The reference exit(1)s on failure (silently for fopen and for the size line, with "Could not
 * process Matrix Market banner." for the banner); this returns BSPGEMM_ERR_IO / _SIZE / _FORMAT
 * so that the CLI can reproduce that behaviour.  Arrays are malloc'd; release with free().     */
bspgemm_status bspgemm_readCOO(const char *path, uint32_t **row, uint32_t **col,
                               uint32_t *M, uint32_t *N, uint32_t *nnz);
/* The same loader with options.  flags = 0 is bspgemm_readCOO.  The
 * reference parses the banner's symmetry token (final/mmio.c:96-179) and then ignores it
 * (final/utils.c:66-71): BSPGEMM_READ_EXPAND_SYMMETRIC mirrors the stored triangle of a
 * symmetric / hermitian / skew-symmetric file so that the CSR holds the full pattern (*nnz is
 * the expanded count).  Off by default: the default result is the reference's, bit for bit.   */
#define BSPGEMM_READ_EXPAND_SYMMETRIC 1u
bspgemm_status bspgemm_readCOO_ex(const char *path, unsigned flags, uint32_t **row, uint32_t **col,
                                  uint32_t *M, uint32_t *N, uint32_t *nnz);
/* writes a CSR as `%%MatrixMarket matrix coordinate pattern general` such that
 * bspgemm_readCOO (and the reference's readCOO) reconstruct exactly this CSR                  */
bspgemm_status bspgemm_write_mtx(const char *path, int rows, int cols,
                                 const int *row_ptr, const int *col_idx);
/* C -> Matrix Market in the FILE's orientation (transposed back), int64 row_ptr              */
bspgemm_status bspgemm_write_result_mtx(const char *path, int rows, int cols,
                                        const int64_t *row_ptr, const int *col_idx);

/* replaces SpGEMM_valid, final/SpGEMM_mpi_omp_validity.c:290-302: exact CSR equality, 1 = same */
int bspgemm_csr_equal(const int *Acol, const int *Arow, const int *Bcol, const int *Brow, int n);
int bspgemm_csr_equal64(const int *Acol, const int64_t *Arow, const int *Bcol, const int64_t *Brow, int n);

/* Seeded synthetic boolean matrices (the reference's inputs came from Matlab
 * sprand via Matlab/write_spm.m:5-8).  Rows sorted, duplicates collapsed.  malloc'd outputs.  */
bspgemm_status bspgemm_gen_uniform(int n, int d, uint64_t seed, int **row_ptr, int **col_idx);
bspgemm_status bspgemm_gen_rmat(int scale, int edge_factor, double a, double b, double c,
                                uint64_t seed, int **row_ptr, int **col_idx);
bspgemm_status bspgemm_gen_powerlaw(int n, int mean_degree, double alpha, int max_degree,
                                    uint64_t seed, int **row_ptr, int **col_idx);

#ifdef __cplusplus
}
#endif
#endif /* BSPGEMM_H */
