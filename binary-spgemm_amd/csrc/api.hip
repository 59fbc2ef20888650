// api.hip -- the C ABI of include/bspgemm.h over the HIP kernels (native handle API, int32
// drop-ins, RCCL stitch).  No CPU compute path exists in this library: without a gfx950 device
// every compute entry point fails with BSPGEMM_ERR_NO_DEVICE.
#include "../../include/bspgemm.h"
#include "kernels.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <sys/mman.h>
#include <mutex>
#include <new>

using namespace bsp;

// ------------------------------------------------------------------ errors ---------------
static thread_local char g_err[512] = "";

static bspgemm_status fail(bspgemm_status st, const char *what, const char *file, int line)
{
    snprintf(g_err, sizeof g_err, "%s (%s:%d)", what, file, line);
    return st;
}
#define FAIL(st, what) fail((st), (what), __FILE__, __LINE__)
#define HIPCHK(call)                                                                       \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            snprintf(g_err, sizeof g_err, "%s: %s (%s:%d)", #call, hipGetErrorString(e_),  \
                     __FILE__, __LINE__);                                                  \
            return (e_ == hipErrorOutOfMemory) ? BSPGEMM_ERR_ALLOC : BSPGEMM_ERR_HIP;      \
        }                                                                                  \
    } while (0)
#define NCCLCHK(call)                                                                      \
    do {                                                                                   \
        ncclResult_t r_ = (call);                                                          \
        if (r_ != ncclSuccess) {                                                           \
            snprintf(g_err, sizeof g_err, "%s: %s (%s:%d)", #call, ncclGetErrorString(r_), \
                     __FILE__, __LINE__);                                                  \
            return BSPGEMM_ERR_COMM;                                                       \
        }                                                                                  \
    } while (0)

extern "C" const char *bspgemm_last_error(void) { return g_err; }

extern "C" const char *bspgemm_status_string(bspgemm_status s)
{
    switch (s) {
    case BSPGEMM_OK: return "ok";
    case BSPGEMM_ERR_INVALID: return "invalid argument";
    case BSPGEMM_ERR_ALLOC: return "allocation failed";
    case BSPGEMM_ERR_HIP: return "HIP runtime error";
    case BSPGEMM_ERR_NO_DEVICE: return "no gfx950 device (this library has no CPU fallback)";
    case BSPGEMM_ERR_OVERFLOW: return "result exceeds the int32 drop-in interface";
    case BSPGEMM_ERR_IO: return "file I/O error";
    case BSPGEMM_ERR_FORMAT: return "Matrix Market format rejected";
    case BSPGEMM_ERR_COMM: return "RCCL error";
    }
    return "unknown status";
}

// ------------------------------------------------------------------ objects --------------
constexpr int kMaxTiles = 16;   // row super-tiles whose compaction overlaps the next tile's accumulate

struct HostScalars {
    long long totalF;
    long long nnzC;
    int bin_count[kNumBins];
    int a_lo, a_hi;
    long long products;                 // true product count of a masked multiply (totalF is the mask total there)
    long long fb[kMaxTiles + 1];        // Fprefix at the super-tile boundaries
};

struct bspgemm_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t stream_b = nullptr;     // second accumulate stream (capacity classes run concurrently)
    hipStream_t stream_c = nullptr;     // compaction stream
    hipEvent_t ev_tile[kMaxTiles][3] = {};          // per super-tile: classes on stream_b done / scan done / start fence
    hipEvent_t ev_cls[kMaxTiles][kNumBins][2] = {}; // per (super-tile, class): launch brackets
    hipEvent_t ev_join = nullptr;
    long long *stitch_partials = nullptr;           // scan scratch of bspgemm_lengths_to_row_ptr
    size_t stitch_partials_cap = 0;
    int *h_bin_tiles = nullptr;         // pinned copy of bin_tiles
    size_t h_bin_tiles_cap = 0;
    // per-row workspace (capacity rows_cap rows)
    size_t rows_cap = 0;
    long long *F = nullptr, *Fprefix = nullptr, *partials = nullptr, *recpre = nullptr, *Fmask = nullptr;
    int *cnt = nullptr, *bin_tiles = nullptr, *bin_count = nullptr;
    RowRec *rec = nullptr;
    // per-A-nonzero workspace: (start,length) of the B row behind every A nonzero
    size_t ab_cap = 0;
    int2 *ab = nullptr;
    // upper-bound placed rows
    size_t tmp_cap = 0;
    int *tmp = nullptr;
    HostScalars *h = nullptr;          // pinned
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // freed result buffers, reused by the next multiply (results are allocated per call like the
    // reference's per-call malloc of Ccol, final/SpGEMM_mpi_omp.c:115, without paying hipMalloc)
    struct CachedBuf { void *p; size_t bytes; };
    CachedBuf cache[8] = {};
    bspgemm_stats stats;
    bool stats_valid = false;
};

extern "C" int bspgemm_par_max_plus_one(const int *idx, long long n);              // host/par_copy.c
extern "C" void bspgemm_par_prefault(void *p, size_t bytes);

struct bspgemm_matrix {
    bspgemm_context *ctx;
    int rows, cols;
    long long nnz;
    int *d_row_ptr, *d_col_idx;
    bool owned;
};

struct bspgemm_result {
    bspgemm_context *ctx;
    int rows;
    long long nnz;
    long long *d_row_ptr;
    int *d_col_idx;
    long long col_cap;      // entries allocated for d_col_idx (upper bound F >= nnz)
};

static bspgemm_status use_device(bspgemm_context *ctx)
{
    HIPCHK(hipSetDevice(ctx->device));
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_create(int device, bspgemm_context **out)
{
    if (!out) return FAIL(BSPGEMM_ERR_INVALID, "ctx is NULL");
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return FAIL(BSPGEMM_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= ndev) return FAIL(BSPGEMM_ERR_INVALID, "device index out of range");
    hipDeviceProp_t prop;
    HIPCHK(hipGetDeviceProperties(&prop, device));
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        snprintf(g_err, sizeof g_err, "device %d is %s; libbspgemm carries gfx950 code only", device,
                 prop.gcnArchName);
        return BSPGEMM_ERR_NO_DEVICE;
    }
    bspgemm_context *ctx = new (std::nothrow) bspgemm_context();
    if (!ctx) return FAIL(BSPGEMM_ERR_ALLOC, "context");
    ctx->device = device;
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
    ctx->own_stream = true;
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&ctx->h), sizeof(HostScalars), hipHostMallocDefault));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->bin_count), kNumBins * sizeof(int)));
    for (auto &e : ctx->ev) HIPCHK(hipEventCreate(&e));
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream_b, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream_c, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    for (auto &t : ctx->ev_tile) for (auto &e : t) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto &t : ctx->ev_cls) for (auto &c : t) for (auto &e : c) HIPCHK(hipEventCreate(&e));
    *out = ctx;
    return BSPGEMM_OK;
}

extern "C" void bspgemm_destroy(bspgemm_context *ctx)
{
    if (!ctx) return;
    hipSetDevice(ctx->device);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    hipFree(ctx->F); hipFree(ctx->Fprefix); hipFree(ctx->partials);
    hipFree(ctx->cnt); hipFree(ctx->bin_tiles); hipFree(ctx->bin_count); hipFree(ctx->tmp);
    hipFree(ctx->rec); hipFree(ctx->recpre); hipFree(ctx->ab); hipFree(ctx->Fmask);
    if (ctx->h) hipHostFree(ctx->h);
    for (auto &e : ctx->ev) if (e) hipEventDestroy(e);
    for (auto &c : ctx->cache) if (c.p) hipFree(c.p);
    if (ctx->stream_b) { hipStreamSynchronize(ctx->stream_b); hipStreamDestroy(ctx->stream_b); }
    if (ctx->stream_c) { hipStreamSynchronize(ctx->stream_c); hipStreamDestroy(ctx->stream_c); }
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    hipFree(ctx->stitch_partials);
    for (auto &t : ctx->ev_tile) for (auto &e : t) if (e) hipEventDestroy(e);
    for (auto &t : ctx->ev_cls) for (auto &c : t) for (auto &e : c) if (e) hipEventDestroy(e);
    if (ctx->h_bin_tiles) hipHostFree(ctx->h_bin_tiles);
    if (ctx->own_stream && ctx->stream) hipStreamDestroy(ctx->stream);
    delete ctx;
}

extern "C" bspgemm_status bspgemm_set_stream(bspgemm_context *ctx, void *hip_stream)
{
    if (!ctx) return FAIL(BSPGEMM_ERR_INVALID, "ctx is NULL");
    if (bspgemm_status st = use_device(ctx)) return st;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    if (hip_stream) {
        if (ctx->own_stream) hipStreamDestroy(ctx->stream);
        ctx->stream = static_cast<hipStream_t>(hip_stream);
        ctx->own_stream = false;
    } else if (!ctx->own_stream) {
        HIPCHK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        ctx->own_stream = true;
    }
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_synchronize(bspgemm_context *ctx)
{
    if (!ctx) return FAIL(BSPGEMM_ERR_INVALID, "ctx is NULL");
    if (bspgemm_status st = use_device(ctx)) return st;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BSPGEMM_OK;
}

// ------------------------------------------------------------------ operands -------------
extern "C" bspgemm_status bspgemm_matrix_upload(bspgemm_context *ctx, int rows, int cols,
                                                const int *row_ptr, const int *col_idx,
                                                bspgemm_matrix **out)
{
    if (!ctx || !out || !row_ptr || rows < 0 || cols < 0) return FAIL(BSPGEMM_ERR_INVALID, "matrix_upload");
    *out = nullptr;
    const long long base = row_ptr[0];
    const long long nnz = (long long)row_ptr[rows] - base;
    if (nnz < 0 || (nnz > 0 && !col_idx)) return FAIL(BSPGEMM_ERR_INVALID, "row_ptr not ascending / col_idx NULL");
    if (bspgemm_status st = use_device(ctx)) return st;
    bspgemm_matrix *m = new (std::nothrow) bspgemm_matrix{ctx, rows, cols, nnz, nullptr, nullptr, true};
    if (!m) return FAIL(BSPGEMM_ERR_ALLOC, "matrix");
    // +1 int of slack on col_idx so an empty matrix still has a valid pointer
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->d_row_ptr), ((size_t)rows + 1) * sizeof(int)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->d_col_idx), ((size_t)nnz + 1) * sizeof(int)));
    HIPCHK(hipMemcpyAsync(m->d_row_ptr, row_ptr, ((size_t)rows + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    if (nnz > 0)
        HIPCHK(hipMemcpyAsync(m->d_col_idx, col_idx + base, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    launch_rebase_i32(m->d_row_ptr, rows + 1, (int)base, ctx->stream);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *out = m;
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_matrix_wrap_device(bspgemm_context *ctx, int rows, int cols, int64_t nnz,
                                                     const int *d_row_ptr, const int *d_col_idx,
                                                     bspgemm_matrix **out)
{
    if (!ctx || !out || !d_row_ptr || rows < 0 || cols < 0 || nnz < 0) return FAIL(BSPGEMM_ERR_INVALID, "matrix_wrap_device");
    bspgemm_matrix *m = new (std::nothrow) bspgemm_matrix{ctx, rows, cols, (long long)nnz,
                                                          const_cast<int *>(d_row_ptr),
                                                          const_cast<int *>(d_col_idx), false};
    if (!m) return FAIL(BSPGEMM_ERR_ALLOC, "matrix");
    *out = m;
    return BSPGEMM_OK;
}

extern "C" void bspgemm_matrix_free(bspgemm_matrix *m)
{
    if (!m) return;
    if (m->owned) {
        hipSetDevice(m->ctx->device);
        hipFree(m->d_row_ptr);
        hipFree(m->d_col_idx);
    }
    delete m;
}
extern "C" int bspgemm_matrix_rows(const bspgemm_matrix *m) { return m ? m->rows : 0; }
extern "C" int bspgemm_matrix_cols(const bspgemm_matrix *m) { return m ? m->cols : 0; }
extern "C" int64_t bspgemm_matrix_nnz(const bspgemm_matrix *m) { return m ? m->nnz : 0; }

// ------------------------------------------------------------------ workspace ------------
static bspgemm_status ensure_rows(bspgemm_context *ctx, size_t rows)
{
    if (rows <= ctx->rows_cap) return BSPGEMM_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    hipFree(ctx->F); hipFree(ctx->Fprefix); hipFree(ctx->partials); hipFree(ctx->cnt); hipFree(ctx->bin_tiles);
    hipFree(ctx->rec); hipFree(ctx->recpre); hipFree(ctx->Fmask);
    ctx->F = ctx->Fprefix = ctx->partials = ctx->recpre = ctx->Fmask = nullptr;
    ctx->cnt = ctx->bin_tiles = nullptr;
    ctx->rec = nullptr;
    ctx->rows_cap = 0;
    const size_t cap = rows + rows / 8 + 64;
    const size_t tiles = cap / 2048 + 2;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->F), cap * sizeof(long long)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->Fmask), cap * sizeof(long long)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->Fprefix), (cap + 1) * sizeof(long long)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->partials), (tiles + 1) * sizeof(long long)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->cnt), cap * sizeof(int)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->bin_tiles), (tiles + 1) * kNumBins * sizeof(int)));
    if (ctx->h_bin_tiles) hipHostFree(ctx->h_bin_tiles);
    ctx->h_bin_tiles = nullptr;
    HIPCHK(hipHostMalloc(reinterpret_cast<void **>(&ctx->h_bin_tiles), (tiles + 1) * kNumBins * sizeof(int), hipHostMallocDefault));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->rec), cap * sizeof(RowRec)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->recpre), cap * sizeof(long long)));
    ctx->rows_cap = cap;
    return BSPGEMM_OK;
}

static bspgemm_status ensure_ab(bspgemm_context *ctx, size_t pairs)
{
    if (pairs <= ctx->ab_cap) return BSPGEMM_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    hipFree(ctx->ab);
    ctx->ab = nullptr;
    ctx->ab_cap = 0;
    const size_t cap = pairs + pairs / 16 + 64;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->ab), cap * sizeof(int2)));
    ctx->ab_cap = cap;
    return BSPGEMM_OK;
}

static bspgemm_status ensure_tmp(bspgemm_context *ctx, size_t ints)
{
    if (ints <= ctx->tmp_cap) return BSPGEMM_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    hipFree(ctx->tmp);
    ctx->tmp = nullptr;
    ctx->tmp_cap = 0;
    const size_t cap = ints + ints / 16 + 1024;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->tmp), cap * sizeof(int)));
    ctx->tmp_cap = cap;
    return BSPGEMM_OK;
}

// result buffers: best fit from the context's cache of freed results, else hipMalloc
static hipError_t result_alloc(bspgemm_context *ctx, void **out, size_t bytes)
{
    int best = -1;
    for (int i = 0; i < 8; i++) {
        const auto &c = ctx->cache[i];
        if (c.p && c.bytes >= bytes && c.bytes <= 2 * bytes + (1 << 20) &&
            (best < 0 || c.bytes < ctx->cache[best].bytes))
            best = i;
    }
    if (best >= 0) {
        *out = ctx->cache[best].p;
        ctx->cache[best].p = nullptr;
        return hipSuccess;
    }
    hipError_t e = hipMalloc(out, bytes);
    if (e == hipErrorOutOfMemory) {                     // drop the cache and retry once
        for (auto &c : ctx->cache) if (c.p) { hipFree(c.p); c.p = nullptr; }
        (void)hipGetLastError();
        e = hipMalloc(out, bytes);
    }
    return e;
}

static size_t result_bytes_rowptr(int rows) { return ((size_t)rows + 1) * sizeof(long long); }
static size_t result_bytes_colidx(long long nnz) { return ((size_t)nnz + 4) * sizeof(int); }

static void result_release(bspgemm_context *ctx, void *p, size_t bytes)
{
    if (!p) return;
    int slot = -1;
    for (int i = 0; i < 8; i++)
        if (!ctx->cache[i].p) { slot = i; break; }
    if (slot < 0) {                                     // evict the smallest cached buffer
        slot = 0;
        for (int i = 1; i < 8; i++)
            if (ctx->cache[i].bytes < ctx->cache[slot].bytes) slot = i;
        if (ctx->cache[slot].bytes >= bytes) { hipFree(p); return; }
        hipFree(ctx->cache[slot].p);
    }
    ctx->cache[slot].p = p;
    ctx->cache[slot].bytes = bytes;
}

// ------------------------------------------------------------------ multiply -------------
static bspgemm_status check_operands(bspgemm_context *ctx, const bspgemm_matrix *A, const bspgemm_matrix *B,
                                     int row_begin, int row_end)
{
    if (!ctx || !A || !B) return FAIL(BSPGEMM_ERR_INVALID, "NULL operand");
    if (A->ctx != ctx || B->ctx != ctx) return FAIL(BSPGEMM_ERR_INVALID, "operand belongs to another context");
    if (row_begin < 0 || row_end < row_begin || row_end > A->rows) return FAIL(BSPGEMM_ERR_INVALID, "row range");
    if (B->rows < A->cols) return FAIL(BSPGEMM_ERR_INVALID, "B has fewer rows than A has columns");
    return BSPGEMM_OK;
}

static bspgemm_status multiply_impl(bspgemm_context *ctx, const bspgemm_matrix *A,
                                    const bspgemm_matrix *B, const bspgemm_matrix *Fm,
                                    int row_begin, int row_end, bspgemm_result **out)
{
    if (!out) return FAIL(BSPGEMM_ERR_INVALID, "result pointer is NULL");
    *out = nullptr;
    if (bspgemm_status st = check_operands(ctx, A, B, row_begin, row_end)) return st;
    if (Fm && (Fm->ctx != ctx || Fm->rows < row_end)) return FAIL(BSPGEMM_ERR_INVALID, "mask has fewer rows than A / wrong context");
    if (bspgemm_status st = use_device(ctx)) return st;
    const int R = row_end - row_begin;
    hipStream_t s = ctx->stream;
    if (bspgemm_status st = ensure_rows(ctx, (size_t)R + 1)) return st;
    if (bspgemm_status st = ensure_ab(ctx, (size_t)A->nnz + 1)) return st;

    bspgemm_result *C = new (std::nothrow) bspgemm_result{ctx, R, 0, nullptr, nullptr, 0};
    if (!C) return FAIL(BSPGEMM_ERR_ALLOC, "result");
    auto bail = [&](bspgemm_status st) { bspgemm_result_free(C); return st; };
#define HIPCHK_C(call)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            snprintf(g_err, sizeof g_err, "%s: %s (%s:%d)", #call, hipGetErrorString(e_),  \
                     __FILE__, __LINE__);                                                  \
            return bail((e_ == hipErrorOutOfMemory) ? BSPGEMM_ERR_ALLOC : BSPGEMM_ERR_HIP);\
        }                                                                                  \
    } while (0)

    hipStream_t sB = ctx->stream_b, sC = ctx->stream_c;
    HIPCHK_C(hipEventRecord(ctx->ev[0], s));
    HIPCHK_C(result_alloc(ctx, reinterpret_cast<void **>(&C->d_row_ptr), result_bytes_rowptr(R)));

    // Row super-tiles (multiples of the 2048-row scan tile).  Tile k's rows are accumulated, their
    // counts scanned, and their compaction runs on its own stream while tile k+1 is accumulated.
    const int scan_tiles = (R + 2047) / 2048;
    int T = 1;                                         // BSPGEMM_TILES > 1 turns the pipelining on
    if (const char *e = getenv("BSPGEMM_TILES")) T = atoi(e);
    if (T < 1) T = 1;
    if (T > kMaxTiles) T = kMaxTiles;
    if (R < (1 << 18)) T = 1;
    if (T > scan_tiles && scan_tiles > 0) T = scan_tiles;
    int tb[kMaxTiles + 1];
    for (int k = 0; k <= T; k++) {
        long long t = (long long)scan_tiles * k / T * 2048;
        tb[k] = (k == T || t > R) ? R : (int)t;
    }
    auto tile_index = [&](int row) { return row >= R ? scan_tiles : row / 2048; };

    // ---- symbolic: per-row products, their prefix, capacity classes ---------------------
    launch_row_work(A->d_row_ptr, A->d_col_idx, B->d_row_ptr, row_begin, row_end, ctx->F, ctx->ab, s);
    HostScalars *h = ctx->h;
    h->products = -1;
    const long long *size_by = ctx->F;          // what rows are binned and offset by: products ...
    if (Fm) {                                   // ... or, masked, the mask row's length (|C_i| <= |F_i|)
        launch_sum_i64(ctx->F, R, ctx->partials, s);
        HIPCHK_C(hipMemcpyAsync(&h->products, ctx->partials + (R > 0 ? (R + 2047) / 2048 : 0), sizeof(long long),
                                hipMemcpyDeviceToHost, s));
        launch_mask_lengths(ctx->F, Fm->d_row_ptr, row_begin, R, ctx->Fmask, s);
        size_by = ctx->Fmask;
    }
    launch_scan_and_bin(size_by, R, row_begin, A->d_row_ptr, ctx->Fprefix, ctx->partials, ctx->bin_tiles,
                        ctx->bin_count, ctx->rec, ctx->recpre, ctx->cnt, s);
    HIPCHK_C(hipMemcpyAsync(&h->totalF, ctx->Fprefix + R, sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK_C(hipMemcpyAsync(h->bin_count, ctx->bin_count, kNumBins * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK_C(hipMemcpyAsync(&h->a_lo, A->d_row_ptr + row_begin, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK_C(hipMemcpyAsync(&h->a_hi, A->d_row_ptr + row_end, sizeof(int), hipMemcpyDeviceToHost, s));
    if (R > 0) {
        HIPCHK_C(hipMemcpyAsync(ctx->h_bin_tiles, ctx->bin_tiles, ((size_t)scan_tiles + 1) * kNumBins * sizeof(int),
                                hipMemcpyDeviceToHost, s));
        for (int k = 0; k <= T; k++)
            HIPCHK_C(hipMemcpyAsync(&h->fb[k], ctx->Fprefix + tb[k], sizeof(long long), hipMemcpyDeviceToHost, s));
    }
    HIPCHK_C(hipEventRecord(ctx->ev[1], s));
    HIPCHK_C(hipStreamSynchronize(s));
    const long long totalF = R > 0 ? h->totalF : 0;
    if (R == 0) memset(h->bin_count, 0, sizeof h->bin_count);
    if (bspgemm_status st = ensure_tmp(ctx, (size_t)totalF + 1)) return bail(st);
    // C.col_idx is taken with the upper-bound size F (known now) so that no size read-back sits
    // between the accumulate and the compaction of a tile; nnz(C) <= F entries of it are used
    HIPCHK_C(result_alloc(ctx, reinterpret_cast<void **>(&C->d_col_idx), result_bytes_colidx(totalF)));
    C->col_cap = totalF;

    // ---- numeric + stitch, pipelined over the super-tiles --------------------------------
    const int levels = wave_levels_for_cols(B->cols);
    size_t bin_start[kNumBins + 1] = {0, 0};               // class b's segment of rec[] (class 0 has none)
    for (int b = 1; b < kNumBins; b++) bin_start[b + 1] = bin_start[b] + (size_t)h->bin_count[b];
    int cls_n[kMaxTiles][kNumBins] = {};
    // The class launches of a tile are independent (disjoint rows of tmp and cnt): they are spread
    // round-robin over two (at most three) streams so that one launch's draining tail overlaps the next
    // launch's ramp-up.  With super-tiles, stream_c is busy compacting the previous tile.
    int class_streams = 2;                             // measured: 2 -8 % numeric time, 3 no better
    if (const char *e = getenv("BSPGEMM_CLASS_STREAMS")) class_streams = atoi(e);
    hipStream_t lanes[3] = {s, sB, sC};
    const int nlanes = T > 1 ? 2 : (class_streams < 1 ? 1 : (class_streams > 3 ? 3 : class_streams));
    for (int k = 0; k < T && R > 0; k++) {
        const int *bt0 = ctx->h_bin_tiles + (size_t)tile_index(tb[k]) * kNumBins;
        const int *bt1 = ctx->h_bin_tiles + (size_t)tile_index(tb[k + 1]) * kNumBins;
        for (int pos = 1; pos < kNumBins; pos++) {
            // launch order: the heavy rows first (few long-running workgroups: started early they
            // finish under the other classes instead of being the multiply's tail), then the
            // one-wave classes by capacity
            const int b = pos == 1 ? kDenseBin : pos - 1;
            const int n = bt1[b] - bt0[b];
            cls_n[k][b] = n;
            if (n <= 0) continue;
            hipStream_t sx = lanes[pos % nlanes];
            const RowRec *rec = ctx->rec + bin_start[b] + bt0[b];
            const long long *recpre = ctx->recpre + bin_start[b] + bt0[b];
            HIPCHK_C(hipEventRecord(ctx->ev_cls[k][b][0], sx));
            if (Fm && b <= kWaveBins && wave_masked_supported(B->cols))
                launch_wave_masked(b, ctx->ab, B->d_col_idx, B->cols, Fm->d_row_ptr, Fm->d_col_idx, rec, recpre, n,
                                   row_begin, ctx->tmp, ctx->cnt, sx);
            else if (Fm)
                HIPCHK_C(launch_dense_rows_masked(ctx->ab, B->d_col_idx, B->cols, rec, recpre, n, row_begin,
                                                  ctx->tmp, ctx->cnt, Fm->d_row_ptr, Fm->d_col_idx, sx));
            else if (b <= kWaveBins)
                launch_wave_rows(b, levels, ctx->ab, B->d_col_idx, B->cols, rec, recpre, n, row_begin,
                                 ctx->tmp, ctx->cnt, sx);
            else
                HIPCHK_C(launch_dense_rows(ctx->ab, B->d_col_idx, B->cols, rec, recpre, n, row_begin,
                                           ctx->tmp, ctx->cnt, sx));
            HIPCHK_C(hipEventRecord(ctx->ev_cls[k][b][1], sx));
        }
        HIPCHK_C(hipGetLastError());
        HIPCHK_C(hipEventRecord(ctx->ev_tile[k][0], sB));
        HIPCHK_C(hipStreamWaitEvent(s, ctx->ev_tile[k][0], 0));
        if (nlanes > 2) {
            HIPCHK_C(hipEventRecord(ctx->ev_tile[k][2], sC));
            HIPCHK_C(hipStreamWaitEvent(s, ctx->ev_tile[k][2], 0));
        }
        // counts -> row_ptr of this tile, continuing from the previous tile's last entry
        launch_scan_counts(ctx->cnt + tb[k], tb[k + 1] - tb[k], C->d_row_ptr + tb[k], ctx->partials,
                           k == 0 ? nullptr : C->d_row_ptr + tb[k], s);
        HIPCHK_C(hipEventRecord(ctx->ev_tile[k][1], s));
        HIPCHK_C(hipStreamWaitEvent(sC, ctx->ev_tile[k][1], 0));
        launch_compact(ctx->tmp, ctx->Fprefix, C->d_row_ptr, tb[k], tb[k + 1], h->fb[k + 1] - h->fb[k],
                       C->d_col_idx, sC);
        HIPCHK_C(hipGetLastError());
    }
    if (R == 0) HIPCHK_C(hipMemsetAsync(C->d_row_ptr, 0, sizeof(long long), s));
    HIPCHK_C(hipEventRecord(ctx->ev[2], s));
    HIPCHK_C(hipEventRecord(ctx->ev_join, sC));
    HIPCHK_C(hipStreamWaitEvent(s, ctx->ev_join, 0));
    HIPCHK_C(hipMemcpyAsync(&h->nnzC, C->d_row_ptr + R, sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK_C(hipEventRecord(ctx->ev[3], s));
    HIPCHK_C(hipStreamSynchronize(s));
    C->nnz = h->nnzC;
#undef HIPCHK_C

    bspgemm_stats &st = ctx->stats;
    memset(&st, 0, sizeof st);
    st.rows = R;
    st.nnz_a = R > 0 ? (long long)h->a_hi - h->a_lo : 0;
    st.products = (Fm && R > 0) ? h->products : totalF;
    st.nnz_c = C->nnz;
    st.bytes_alg = 4ll * (R + 1) + 12ll * st.nnz_a + 4ll * st.products + 4ll * C->nnz + 8ll * (R + 1);
    static_assert(kMaxBins == BSPGEMM_MAX_BINS && kNumBins <= kMaxBins, "stats arrays hold every class");
    st.bins = kNumBins;
    for (int b = 0; b < kNumBins; b++) {
        st.rows_per_bin[b] = h->bin_count[b];
        st.bin_cap[b] = b == 0 ? 0 : (b == kDenseBin ? 0x7fffffff : 64 * kWaveChunks[b]);
    }
    hipEventElapsedTime(&st.ms_total, ctx->ev[0], ctx->ev[3]);
    hipEventElapsedTime(&st.ms_symbolic, ctx->ev[0], ctx->ev[1]);
    hipEventElapsedTime(&st.ms_numeric, ctx->ev[1], ctx->ev[2]);
    hipEventElapsedTime(&st.ms_stitch, ctx->ev[2], ctx->ev[3]);
    st.tiles = T;
    for (int k = 0; k < T; k++)
        for (int b = 1; b < kNumBins; b++)
            if (cls_n[k][b] > 0) {
                float ms = 0;
                hipEventElapsedTime(&ms, ctx->ev_cls[k][b][0], ctx->ev_cls[k][b][1]);
                st.ms_bin[b] += ms;
            }
    ctx->stats_valid = true;
    *out = C;
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_multiply(bspgemm_context *ctx, const bspgemm_matrix *A,
                                           const bspgemm_matrix *B, int row_begin, int row_end,
                                           bspgemm_result **out)
{
    return multiply_impl(ctx, A, B, nullptr, row_begin, row_end, out);
}

// First implementation of the masked product: every non-empty row goes through the dense-window
// kernel with a second (kept-bits) bitmap.  Correct for any shape; a mask-first rank-bitmap path
// for short rows is the planned fast path.
extern "C" bspgemm_status bspgemm_multiply_masked(bspgemm_context *ctx, const bspgemm_matrix *A,
                                                  const bspgemm_matrix *B, const bspgemm_matrix *F,
                                                  int row_begin, int row_end, bspgemm_result **out)
{
    if (!F) {
        if (out) *out = nullptr;
        return FAIL(BSPGEMM_ERR_INVALID, "mask is NULL");
    }
    return multiply_impl(ctx, A, B, F, row_begin, row_end, out);
}

extern "C" int bspgemm_result_rows(const bspgemm_result *C) { return C ? C->rows : 0; }
extern "C" int64_t bspgemm_result_nnz(const bspgemm_result *C) { return C ? C->nnz : 0; }
extern "C" const int64_t *bspgemm_result_row_ptr_device(const bspgemm_result *C)
{
    return C ? reinterpret_cast<const int64_t *>(C->d_row_ptr) : nullptr;
}
extern "C" const int *bspgemm_result_col_idx_device(const bspgemm_result *C) { return C ? C->d_col_idx : nullptr; }

extern "C" bspgemm_status bspgemm_result_download(bspgemm_context *ctx, const bspgemm_result *C,
                                                  int64_t *row_ptr, int *col_idx)
{
    if (!ctx || !C) return FAIL(BSPGEMM_ERR_INVALID, "result_download");
    if (bspgemm_status st = use_device(ctx)) return st;
    if (row_ptr)
        HIPCHK(hipMemcpyAsync(row_ptr, C->d_row_ptr, ((size_t)C->rows + 1) * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    // (a pinned-staging pipeline with OpenMP copies out of it was measured 3x SLOWER than this
    // plain pageable copy for a 5.3 GB result: user-space first-touch faults of the fresh
    // destination cost more than the runtime's in-kernel pinning of the same pages)
    if (col_idx && C->nnz > 0)
        HIPCHK(hipMemcpyAsync(col_idx, C->d_col_idx, (size_t)C->nnz * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BSPGEMM_OK;
}

extern "C" void bspgemm_result_free(bspgemm_result *C)
{
    if (!C) return;
    hipSetDevice(C->ctx->device);
    result_release(C->ctx, C->d_row_ptr, result_bytes_rowptr(C->rows));
    result_release(C->ctx, C->d_col_idx, result_bytes_colidx(C->col_cap));
    delete C;
}

extern "C" bspgemm_status bspgemm_last_stats(const bspgemm_context *ctx, bspgemm_stats *out)
{
    if (!ctx || !out || !ctx->stats_valid) return FAIL(BSPGEMM_ERR_INVALID, "no multiply has run on this context");
    *out = ctx->stats;
    return BSPGEMM_OK;
}

// --------------------------------------------------------------- gathered lengths -> row_ptr -
extern "C" bspgemm_status bspgemm_lengths_to_row_ptr(bspgemm_context *ctx, const int *d_lengths, int nranks, int width,
                                                     const int *bounds, int64_t *d_row_ptr, void *hip_stream)
{
    if (!ctx || !d_lengths || !bounds || !d_row_ptr || nranks < 1 || width < 0 || bounds[0] != 0)
        return FAIL(BSPGEMM_ERR_INVALID, "lengths_to_row_ptr");
    for (int r = 0; r < nranks; r++)
        if (bounds[r + 1] < bounds[r] || bounds[r + 1] - bounds[r] > width) return FAIL(BSPGEMM_ERR_INVALID, "bounds");
    if (bspgemm_status st = use_device(ctx)) return st;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);      // NULL is HIP's default stream, as for any launch
    // own scan scratch: this may run on another stream than a multiply that is using ctx->partials
    const size_t need = (size_t)width / 2048 + 2;
    if (need > ctx->stitch_partials_cap) {
        if (ctx->stitch_partials) HIPCHK(hipFree(ctx->stitch_partials));
        ctx->stitch_partials = nullptr;
        ctx->stitch_partials_cap = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->stitch_partials), need * sizeof(long long)));
        ctx->stitch_partials_cap = need;
    }
    long long *out = reinterpret_cast<long long *>(d_row_ptr);
    for (int r = 0; r < nranks; r++)          // shard r continues the row_ptr where shard r-1 ended
        launch_scan_counts(d_lengths + (size_t)r * width, bounds[r + 1] - bounds[r], out + bounds[r],
                           ctx->stitch_partials, r == 0 ? nullptr : out + bounds[r], s);
    HIPCHK(hipGetLastError());
    return BSPGEMM_OK;
}

// ------------------------------------------------------------------ products as operands -
extern "C" bspgemm_status bspgemm_matrix_from_result(bspgemm_context *ctx, const bspgemm_result *C, int cols,
                                                     bspgemm_matrix **out)
{
    if (!ctx || !C || !out || cols < 0 || C->ctx != ctx) return FAIL(BSPGEMM_ERR_INVALID, "matrix_from_result");
    *out = nullptr;
    if (C->nnz > INT_MAX) return FAIL(BSPGEMM_ERR_OVERFLOW, "product has more than INT_MAX nonzeros: not usable as an int32 operand");
    if (bspgemm_status st = use_device(ctx)) return st;
    bspgemm_matrix *m = new (std::nothrow) bspgemm_matrix{ctx, C->rows, cols, C->nnz, nullptr, nullptr, true};
    if (!m) return FAIL(BSPGEMM_ERR_ALLOC, "matrix");
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->d_row_ptr), ((size_t)C->rows + 1) * sizeof(int)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->d_col_idx), ((size_t)C->nnz + 1) * sizeof(int)));
    launch_narrow_row_ptr(C->d_row_ptr, m->d_row_ptr, C->rows + 1, ctx->stream);
    if (C->nnz > 0)
        HIPCHK(hipMemcpyAsync(m->d_col_idx, C->d_col_idx, (size_t)C->nnz * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    *out = m;
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_closure(bspgemm_context *ctx, const bspgemm_matrix *A, int max_iter,
                                          bspgemm_result **T, int *iterations)
{
    if (!ctx || !A || !T || A->ctx != ctx || A->rows != A->cols) return FAIL(BSPGEMM_ERR_INVALID, "closure needs a square matrix");
    *T = nullptr;
    if (iterations) *iterations = 0;
    if (max_iter < 1) max_iter = 1;
    if (A->nnz + (long long)A->rows > INT_MAX) return FAIL(BSPGEMM_ERR_OVERFLOW, "A or I exceeds int32 nonzeros");
    if (bspgemm_status st = use_device(ctx)) return st;
    const int n = A->rows;
    bspgemm_matrix *cur = new (std::nothrow) bspgemm_matrix{ctx, n, n, A->nnz + n, nullptr, nullptr, true};
    if (!cur) return FAIL(BSPGEMM_ERR_ALLOC, "matrix");
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&cur->d_row_ptr), ((size_t)n + 1) * sizeof(int)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&cur->d_col_idx), ((size_t)cur->nnz + 1) * sizeof(int)));
    launch_add_diagonal(A->d_row_ptr, A->d_col_idx, n, cur->d_row_ptr, cur->d_col_idx, ctx->stream);
    HIPCHK(hipStreamSynchronize(ctx->stream));
    long long prev_nnz = -1;      // nnz of the deduplicated T(k); unknown for T0 (may hold duplicates)
    bspgemm_result *C = nullptr;
    bspgemm_status st = BSPGEMM_OK;
    for (int it = 0; it < max_iter; it++) {
        bspgemm_result *next = nullptr;
        st = bspgemm_multiply(ctx, cur, cur, 0, n, &next);
        if (st) break;
        if (iterations) *iterations = it + 1;
        bspgemm_result_free(C);
        C = next;
        if (C->nnz == prev_nnz) break;                  // T*T == T: fixpoint (T contains I, so T <= T*T)
        prev_nnz = C->nnz;
        if (it + 1 == max_iter) break;
        bspgemm_matrix *nm = nullptr;
        st = bspgemm_matrix_from_result(ctx, C, n, &nm);
        if (st) break;
        bspgemm_matrix_free(cur);
        cur = nm;
    }
    bspgemm_matrix_free(cur);
    if (st) { bspgemm_result_free(C); return st; }
    *T = C;
    return BSPGEMM_OK;
}

// ------------------------------------------------------------------ sharding helper ------
extern "C" bspgemm_status bspgemm_row_work_prefix(bspgemm_context *ctx, const bspgemm_matrix *A,
                                                  const bspgemm_matrix *B, int64_t *prefix_host)
{
    if (!prefix_host) return FAIL(BSPGEMM_ERR_INVALID, "prefix_host is NULL");
    if (bspgemm_status st = check_operands(ctx, A, B, 0, A ? A->rows : 0)) return st;
    if (bspgemm_status st = use_device(ctx)) return st;
    const int R = A->rows;
    if (bspgemm_status st = ensure_rows(ctx, (size_t)R + 1)) return st;
    if (bspgemm_status st = ensure_ab(ctx, (size_t)A->nnz + 1)) return st;
    launch_row_work(A->d_row_ptr, A->d_col_idx, B->d_row_ptr, 0, R, ctx->F, ctx->ab, ctx->stream);
    launch_scan_and_bin(ctx->F, R, 0, A->d_row_ptr, ctx->Fprefix, ctx->partials, ctx->bin_tiles, ctx->bin_count,
                        ctx->rec, ctx->recpre, ctx->cnt, ctx->stream);
    HIPCHK(hipMemcpyAsync(prefix_host, ctx->Fprefix, ((size_t)R + 1) * sizeof(long long), hipMemcpyDeviceToHost, ctx->stream));
    HIPCHK(hipStreamSynchronize(ctx->stream));
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_partition_rows(bspgemm_context *ctx, const bspgemm_matrix *A,
                                                 const bspgemm_matrix *B, int parts, int *bounds)
{
    if (!bounds || parts <= 0 || !A) return FAIL(BSPGEMM_ERR_INVALID, "partition_rows");
    const int R = A->rows;
    int64_t *prefix = static_cast<int64_t *>(malloc(((size_t)R + 1) * sizeof(int64_t)));
    if (!prefix) return FAIL(BSPGEMM_ERR_ALLOC, "prefix");
    bspgemm_status st = bspgemm_row_work_prefix(ctx, A, B, prefix);
    if (st == BSPGEMM_OK) {
        // cost of a row = its products + a constant for the per-row overhead
        const long long per_row = 32;
        const long long total = prefix[R] + per_row * R;
        bounds[0] = 0;
        int r = 0;
        for (int p = 1; p < parts; p++) {
            const long long target = total / parts * p;
            while (r < R && prefix[r] + per_row * r < target) r++;
            bounds[p] = r;
        }
        bounds[parts] = R;
    }
    free(prefix);
    return st;
}

// ------------------------------------------------------------------ int32 drop-ins -------
static std::mutex g_dropin_mu;
static bspgemm_context *g_dropin_ctx = nullptr;
static int g_dropin_device = -1;

extern "C" int bspgemm_dropin_set_device(int device)
{
    std::lock_guard<std::mutex> lk(g_dropin_mu);
    if (g_dropin_ctx && g_dropin_device != device) {
        bspgemm_destroy(g_dropin_ctx);
        g_dropin_ctx = nullptr;
    }
    g_dropin_device = device;
    return BSPGEMM_OK;
}

static bspgemm_status dropin_ctx(bspgemm_context **out)
{
    if (!g_dropin_ctx) {
        int dev = g_dropin_device;
        if (dev < 0) {
            const char *e = getenv("BSPGEMM_DEVICE");
            dev = e ? atoi(e) : 0;
        }
        bspgemm_status st = bspgemm_create(dev, &g_dropin_ctx);
        if (st) return st;
        g_dropin_device = dev;
    }
    *out = g_dropin_ctx;
    return BSPGEMM_OK;
}

static int dropin_fail(const char *fn, bspgemm_status st)
{
    fprintf(stderr, "%s: %s: %s\n", fn, bspgemm_status_string(st), bspgemm_last_error());
    return (int)st;
}

// Shared body: C rows [r0,r1) of A*B with host int32 arrays in the reference's conventions.
// mode 0: *Ccol = malloc(nnz) (SpGEMM_omp :115)   mode 1: grow caller's buffer (bigslice :28-31)
// mode 2: caller's buffer is exact (SpGEMM_mat)
// a multi-GB destination would be faulted in page by page inside the device-to-host copy: ask
// for transparent huge pages on its page-aligned interior (no effect where THP is off) and touch
// it from all host threads first (measured on a 5.3 GB result: download 425 -> 320 ms with the
// advice alone)
static void advise_huge(void *p, size_t bytes)
{
    if (!p || bytes < ((size_t)64 << 20)) return;
    const uintptr_t lo = (reinterpret_cast<uintptr_t>(p) + ((size_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1);
    const uintptr_t hi = (reinterpret_cast<uintptr_t>(p) + bytes) & ~(((uintptr_t)2 << 20) - 1);
    if (hi > lo) madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
    bspgemm_par_prefault(p, bytes);      // ... and take the faults (page zeroing) on all host threads
}

static bspgemm_status dropin_run(const int *Acol, const int *Arow, int r0, int r1,
                                 const int *Bcol, const int *Brow, int Bm,
                                 int **Ccol, int *Crow, int *Csize, int mode,
                                 const int *Fcol = nullptr, const int *Frow = nullptr)
{
    if (!Acol || !Arow || !Bcol || !Brow || !Crow || !Ccol || r0 < 0 || r1 < r0 || Bm < 0)
        return FAIL(BSPGEMM_ERR_INVALID, "drop-in arguments");
    std::lock_guard<std::mutex> lk(g_dropin_mu);
    bspgemm_context *ctx;
    if (bspgemm_status st = dropin_ctx(&ctx)) return st;
    const int rows = r1 - r0;
    const bool timing = getenv("BSPGEMM_DROPIN_TIMING") != nullptr;     // stage times to stderr
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    // B's row count is implicit in the reference (never passed): 1 + the largest column of A used
    const int brows = bspgemm_par_max_plus_one(Acol + Arow[r0], (long long)Arow[r1] - Arow[r0]);
    const double t1 = now();
    bspgemm_matrix *A = nullptr, *B = nullptr, *Fm = nullptr;
    bspgemm_result *C = nullptr;
    bspgemm_status st = bspgemm_matrix_upload(ctx, rows, brows, Arow + r0, Acol, &A);
    if (!st) st = bspgemm_matrix_upload(ctx, brows, Bm, Brow, Bcol, &B);
    if (!st && Frow) st = bspgemm_matrix_upload(ctx, rows, Bm, Frow + r0, Fcol, &Fm);
    const double t2 = now();
    if (!st) st = Fm ? bspgemm_multiply_masked(ctx, A, B, Fm, 0, rows, &C) : bspgemm_multiply(ctx, A, B, 0, rows, &C);
    const double t3 = now();
    if (!st) {
        const long long nnz = bspgemm_result_nnz(C);
        if (nnz > INT_MAX) {
            st = FAIL(BSPGEMM_ERR_OVERFLOW, "nnz(C) > INT_MAX: use the int64 handle API");
        } else {
            int *dst = nullptr;
            if (mode == 0) {
                dst = static_cast<int *>(malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int)));
                advise_huge(dst, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int));
            } else if (mode == 1) {
                dst = *Ccol;
                if (!dst || !Csize || *Csize < nnz) {
                    dst = static_cast<int *>(realloc(*Ccol, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int)));
                    if (dst && Csize) *Csize = (int)(nnz > 0 ? nnz : 1);
                    advise_huge(dst, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int));
                }
            } else {
                dst = *Ccol;
            }
            int64_t *rp64 = static_cast<int64_t *>(malloc(((size_t)rows + 1) * sizeof(int64_t)));
            if (!dst || !rp64) {
                st = FAIL(BSPGEMM_ERR_ALLOC, "host result");
                if (mode == 0) free(dst);
            } else {
                st = bspgemm_result_download(ctx, C, rp64, dst);
                if (!st) {
                    for (int i = 0; i <= rows; i++) Crow[i] = (int)rp64[i];
                    *Ccol = dst;
                } else if (mode == 0) {
                    free(dst);
                }
            }
            free(rp64);
        }
    }
    const double t4 = now();
    bspgemm_result_free(C);
    bspgemm_matrix_free(A);
    bspgemm_matrix_free(B);
    bspgemm_matrix_free(Fm);
    if (timing)
        fprintf(stderr, "[bspgemm drop-in] scan A %.1f ms, upload %.1f, multiply %.1f, download %.1f, free %.1f\n",
                t1 - t0, t2 - t1, t3 - t2, t4 - t3, now() - t4);
    return st;
}

extern "C" int SpGEMM_hip(int *Acol, int *Arow, int An, int *Bcol, int *Brow, int Bm,
                          int **Ccol, int *Crow, int tBlock)
{
    (void)tBlock;
    if (Ccol) *Ccol = nullptr;
    bspgemm_status st = dropin_run(Acol, Arow, 0, An, Bcol, Brow, Bm, Ccol, Crow, nullptr, 0);
    return st ? dropin_fail("SpGEMM_hip", st) : 0;
}

extern "C" int SpGEMM_hip_bigslice(int *Acol, int *Arow, int An, int *Bcol, int *Brow, int Bm,
                                   int **Ccol, int *Crow, int *Csize, int start_row, int end_row)
{
    (void)An;
    bspgemm_status st = dropin_run(Acol, Arow, start_row, end_row, Bcol, Brow, Bm, Ccol, Crow, Csize, 1);
    return st ? dropin_fail("SpGEMM_hip_bigslice", st) : 0;
}

extern "C" int SpGEMM_hip_mat(int *Acol, int *Arow, int An, int *Bcol, int *Brow, int Bm,
                              int *Ccol, int *Crow)
{
    int *p = Ccol;
    bspgemm_status st = dropin_run(Acol, Arow, 0, An, Bcol, Brow, Bm, &p, Crow, nullptr, 2);
    return st ? dropin_fail("SpGEMM_hip_mat", st) : 0;
}

extern "C" int SpGEMM_hip_masked(int *Acol, int *Arow, int An, int *Bcol, int *Brow, int Bm,
                                 int *Fcol, int *Frow, int **Ccol, int *Crow, int *Csize)
{
    if (!Fcol || !Frow) return dropin_fail("SpGEMM_hip_masked", FAIL(BSPGEMM_ERR_INVALID, "mask is NULL"));
    bspgemm_status st = dropin_run(Acol, Arow, 0, An, Bcol, Brow, Bm, Ccol, Crow, Csize, 1, Fcol, Frow);
    return st ? dropin_fail("SpGEMM_hip_masked", st) : 0;
}

// ------------------------------------------------------------------ multi-GPU stitch -----
struct bspgemm_comm {
    bspgemm_context *ctx;
    ncclComm_t comm;
    int rank, nranks;
    long long *d_nnz;      // nranks
    int *d_bounds;         // nranks+1
    long long *d_global;   // stitched row_ptr, grown on demand
    size_t global_cap;
};

extern "C" bspgemm_status bspgemm_comm_unique_id(unsigned char id[BSPGEMM_UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == BSPGEMM_UNIQUE_ID_BYTES, "RCCL unique id size");
    if (!id) return FAIL(BSPGEMM_ERR_INVALID, "id is NULL");
    ncclUniqueId u;
    NCCLCHK(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_comm_create(bspgemm_context *ctx, const unsigned char id[BSPGEMM_UNIQUE_ID_BYTES],
                                              int rank, int nranks, bspgemm_comm **out)
{
    if (!ctx || !id || !out || nranks <= 0 || rank < 0 || rank >= nranks) return FAIL(BSPGEMM_ERR_INVALID, "comm_create");
    *out = nullptr;
    if (bspgemm_status st = use_device(ctx)) return st;
    bspgemm_comm *c = new (std::nothrow) bspgemm_comm{ctx, nullptr, rank, nranks, nullptr, nullptr, nullptr, 0};
    if (!c) return FAIL(BSPGEMM_ERR_ALLOC, "comm");
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    NCCLCHK(ncclCommInitRank(&c->comm, nranks, u, rank));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_nnz), (size_t)nranks * sizeof(long long)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_bounds), ((size_t)nranks + 1) * sizeof(int)));
    *out = c;
    return BSPGEMM_OK;
}

extern "C" void bspgemm_comm_destroy(bspgemm_comm *c)
{
    if (!c) return;
    hipSetDevice(c->ctx->device);
    if (c->comm) ncclCommDestroy(c->comm);
    hipFree(c->d_nnz);
    hipFree(c->d_bounds);
    hipFree(c->d_global);
    delete c;
}

// global[i] += sum of shard nnz before the shard that owns row i; global[total_rows] = grand total
__global__ void k_stitch_rebase(long long *global, const int *bounds, const long long *shard_nnz, int nranks)
{
    const int total_rows = bounds[nranks];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i > total_rows) return;
    long long base = 0;
    int r = 0;
    while (r < nranks && i >= bounds[r + 1]) { base += shard_nnz[r]; r++; }
    if (i == total_rows) global[i] = base;
    else global[i] += base;
}

extern "C" bspgemm_status bspgemm_comm_stitch_row_ptr(bspgemm_comm *c, const bspgemm_result *local,
                                                      const int *bounds, const int64_t **d_row_ptr_global,
                                                      int64_t *shard_nnz)
{
    if (!c || !local || !bounds || !d_row_ptr_global) return FAIL(BSPGEMM_ERR_INVALID, "stitch_row_ptr");
    bspgemm_context *ctx = c->ctx;
    if (bspgemm_status st = use_device(ctx)) return st;
    const size_t need = (size_t)bounds[c->nranks] + 1;
    if (need > c->global_cap) {
        HIPCHK(hipStreamSynchronize(ctx->stream));
        hipFree(c->d_global);
        c->d_global = nullptr;
        c->global_cap = 0;
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_global), need * sizeof(long long)));
        c->global_cap = need;
    }
    long long *global = c->d_global;
    *d_row_ptr_global = reinterpret_cast<const int64_t *>(global);
    const int my_rows = bounds[c->rank + 1] - bounds[c->rank];
    if (my_rows != local->rows) return FAIL(BSPGEMM_ERR_INVALID, "local result does not match bounds[rank]");
    hipStream_t s = ctx->stream;
    HIPCHK(hipMemcpyAsync(c->d_bounds, bounds, ((size_t)c->nranks + 1) * sizeof(int), hipMemcpyHostToDevice, s));
    // shard sizes: every rank contributes row_ptr[rows] (its nnz), 8 bytes
    NCCLCHK(ncclAllGather(local->d_row_ptr + local->rows, c->d_nnz, 1, ncclInt64, c->comm, s));
    // row_ptr shards have different lengths (equal-work cuts), so one broadcast per shard, grouped
    NCCLCHK(ncclGroupStart());
    for (int r = 0; r < c->nranks; r++) {
        const int n = bounds[r + 1] - bounds[r];
        if (n <= 0) continue;
        long long *dst = global + bounds[r];
        const void *src = (r == c->rank) ? static_cast<const void *>(local->d_row_ptr) : static_cast<const void *>(dst);
        NCCLCHK(ncclBroadcast(src, dst, (size_t)n, ncclInt64, r, c->comm, s));
    }
    NCCLCHK(ncclGroupEnd());
    const int total_rows = bounds[c->nranks];
    hipLaunchKernelGGL(k_stitch_rebase, dim3((total_rows + 1 + 255) / 256), dim3(256), 0, s,
                       global, c->d_bounds, c->d_nnz, c->nranks);
    if (shard_nnz)
        HIPCHK(hipMemcpyAsync(shard_nnz, c->d_nnz, (size_t)c->nranks * sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return BSPGEMM_OK;
}
