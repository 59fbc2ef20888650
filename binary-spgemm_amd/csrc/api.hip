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
#include <unistd.h>
#include <mutex>
#include <thread>
#include <atomic>
#include <vector>
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
// the same with a clean-up: `bail(status)` must be in scope (frees what the function has built so far)
#define HIPCHK_B(call)                                                                     \
    do {                                                                                   \
        hipError_t e_ = (call);                                                            \
        if (e_ != hipSuccess) {                                                            \
            snprintf(g_err, sizeof g_err, "%s: %s (%s:%d)", #call, hipGetErrorString(e_),  \
                     __FILE__, __LINE__);                                                  \
            return bail((e_ == hipErrorOutOfMemory) ? BSPGEMM_ERR_ALLOC : BSPGEMM_ERR_HIP);\
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
    case BSPGEMM_ERR_FORMAT: return "Matrix Market banner rejected";
    case BSPGEMM_ERR_SIZE: return "Matrix Market size line or entry rejected";
    case BSPGEMM_ERR_COMM: return "RCCL error";
    }
    return "unknown status";
}

// ------------------------------------------------------------------ objects --------------
struct HostScalars {
    long long totalF;
    long long nnzC;
    int bin_count[kNumBins];
    int a_lo, a_hi;
    long long products;                 // true product count of a masked multiply (totalF is the mask total there)
    long long heavy_total;              // entries the heavy rows may need in the workspace
    long long pack_totals[2];           // fused flow: number of tiles, sum of min(F_i, cols)
    unsigned chain_err;                 // fused flow: a look-back wait ran out
    PrepScalars prep;                   // upper-bound flow: the prepass results, fetched in one copy
};

struct bspgemm_context {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    hipStream_t stream_b = nullptr;     // second accumulate stream (capacity classes run concurrently)
    hipStream_t stream_c = nullptr;     // compaction stream
    hipEvent_t ev_tile[2][3] = {};                  // per phase (symbolic count, numeric): fork / joins of the side streams
    hipEvent_t ev_join = nullptr;
    // timing events and counters of the last kStatSlots multiplies (bspgemm_stats_at): a caller that
    // times K steps reads K sets of HIP-event brackets afterwards instead of one
    struct StatSlot {
        hipEvent_t ev[5] = {};                      // start / classes known / row sizes known / rows emitted / done
        hipEvent_t ev_cls[2][kNumBins][2] = {};     // per (phase, class): launch brackets
        int R = 0;
        bool cls_timed = false;                     // the class brackets of this multiply were recorded
        int mid_cap = 0;
        HostScalars h = {};
        long long products = 0, nnz_c = 0;
        int cls_n[2][kNumBins] = {};
        bool used = false;
    };
    static constexpr int kStatSlots = 16;
    StatSlot slots[kStatSlots];
    int slot_head = 0;                              // the most recent multiply's slot
    long long *stitch_partials = nullptr;           // scan scratch of bspgemm_lengths_to_row_ptr
    size_t stitch_partials_cap = 0;
    // per-row workspace (capacity rows_cap rows)
    size_t rows_cap = 0;
    long long *F = nullptr, *Fprefix = nullptr, *partials = nullptr, *recpre = nullptr, *Fmask = nullptr;
    long long *hpartials = nullptr;     // per scan tile: workspace entries of its heavy rows (scanned)
    RowRec *hub_rec = nullptr;          // the hub rows' records by decreasing products (kHeavySortMax entries)
    long long *hub_pre = nullptr;
    int *cnt = nullptr, *bin_tiles = nullptr, *bin_count = nullptr;
    RowRec *rec = nullptr;
    // per-A-nonzero workspace: (start,length) of the B row behind every A nonzero
    size_t ab_cap = 0;
    int2 *ab = nullptr;
    // upper-bound placed rows: the heavy rows of a plain product, every row of a masked one
    size_t tmp_cap = 0;
    int *tmp = nullptr;
    // fused flow (csrc/tile_rows.inc): tile descriptors, look-back chain, packer scratch
    size_t fused_cap = 0;               // rows these are sized for
    TileDesc *tiles = nullptr;
    unsigned long long *chain = nullptr;
    unsigned char *marks8 = nullptr;
    int *tile_count = nullptr;
    long long *tile_bound = nullptr;
    long long *pack_totals = nullptr;   // 2
    PrepScalars *d_prep = nullptr;      // device side of HostScalars::prep
    int *chunk_row = nullptr;           // compaction: row of every kCompactGran-th output (left by the count scan)
    size_t chunk_cap = 0;
    unsigned *tickets = nullptr;        // 8 counters 32 words apart, then the error word
    HostScalars *h = nullptr;          // pinned
    // freed result buffers, reused by the next multiply (results are allocated per call like the
    // reference's per-call malloc of Ccol, final/SpGEMM_mpi_omp.c:115, without paying hipMalloc)
    struct CachedBuf { void *p; size_t bytes; };
    CachedBuf cache[8] = {};
    size_t cache_budget = 0;            // bytes the cache may pin (a quarter of the device memory)
    int flow = BSPGEMM_FLOW_AUTO;       // bspgemm_set_flow / BSPGEMM_FLOW
    // environment knobs, read ONCE in bspgemm_create (include/bspgemm.h, "Environment")
    bool class_timing = false;          // bspgemm_set_class_timing / BSPGEMM_CLASS_TIMING=1: an event pair around every class launch
    int class_streams = 2;              // BSPGEMM_CLASS_STREAMS: streams the class launches alternate over (measured: 2 -8 %, 3 no better)
    bool check = false;                 // BSPGEMM_CHECK: the exact flow never emits on unverified sizes
    int rw_blk = -1;                    // BSPGEMM_RW_BLK: 0 never / 1 always use the blocked extents table (default: per operand)
    bool debug_alloc = false;           // BSPGEMM_DEBUG_ALLOC: allocation trace on stderr
    bool dropin_timing = false;         // BSPGEMM_DROPIN_TIMING: stage times of the int32 drop-ins on stderr
};

extern "C" int bspgemm_par_max_plus_one(const int *idx, long long n);              // host/par_copy.c
extern "C" void bspgemm_par_prefault(void *p, size_t bytes);
extern "C" long long bspgemm_par_output_bound(const int *Acol, const int *Arow, int r0, int r1, const int *Brow, int brows, long long cap);

struct bspgemm_matrix {
    bspgemm_context *ctx;
    int rows, cols;
    long long nnz;
    int *d_row_ptr, *d_col_idx;
    bool owned;
    // row lengths clamped to 255, one byte per row: what a product with this matrix as B gathers
    // per A-nonzero to size its rows (csrc/prepass.hip: k_row_products).  Part of the operand's
    // device layout: built when the operand is created (lazily for wrapped device arrays).
    mutable unsigned char *d_deg8 = nullptr;
    // blocked extents table {row_ptr of every 8th row, 8 clamped lengths}: what k_row_work gathers per
    // A-nonzero instead of a B.row_ptr pair (csrc/prepass.hip: k_row_work_blk); built on first use as B
    mutable int *d_blk8 = nullptr;
    mutable int blk8_state = 0;          // 0 undecided, 1 in use, 2 not worth it for this operand
};

static bspgemm_status ensure_deg8(const bspgemm_matrix *m)
{
    if (m->d_deg8) return BSPGEMM_OK;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->d_deg8), (size_t)m->rows + 1));
    launch_deg8(m->d_row_ptr, m->rows, m->d_deg8, m->ctx->stream);
    HIPCHK(hipGetLastError());
    return BSPGEMM_OK;
}

// Whether products with `m` as B go through the blocked table.  It pays when B.row_ptr is several
// times an XCD's 4 MB L2 (R-MAT scale 22: 16.8 MB, k_row_work 1.20 -> 0.92 ms) and the operand is not
// dominated by rows of 255+ nonzeros, whose lengths the table clamps (power-law n = 2^20: B.row_ptr
// fits L2 anyway and 10 % of the lookups fall through: 2.0 -> 3.0 ms; Graph500 skew: 70 % fall through).
// Decided once per operand: one 8-byte read-back when the table is built.
static bspgemm_status ensure_blk8(const bspgemm_matrix *m)
{
    if (m->blk8_state) return BSPGEMM_OK;
    m->blk8_state = 2;
    const int force = m->ctx->rw_blk;                              // 0 never, 1 always, -1 decide per operand
    if (force == 0 || (force < 0 && m->rows < (1 << 21))) return BSPGEMM_OK;
    const size_t ints = (size_t)3 * (((size_t)m->rows + 7) / 8 + 1);
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->d_blk8), (ints + 4) * sizeof(int)));
    unsigned long long *d_clamped = reinterpret_cast<unsigned long long *>(m->d_blk8 + ((ints + 1) & ~(size_t)1));
    HIPCHK(hipMemsetAsync(d_clamped, 0, sizeof(unsigned long long), m->ctx->stream));
    launch_blk8(m->d_row_ptr, m->rows, m->d_blk8, d_clamped, m->ctx->stream);
    HIPCHK(hipGetLastError());
    unsigned long long clamped = 0;
    HIPCHK(hipMemcpyAsync(&clamped, d_clamped, sizeof(clamped), hipMemcpyDeviceToHost, m->ctx->stream));
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
    if (force == 1 || clamped * 8ull <= (unsigned long long)m->nnz) m->blk8_state = 1;
    return BSPGEMM_OK;
}

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

extern "C" int bspgemm_device_count(void)
{
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return ndev;
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
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->d_prep), sizeof(PrepScalars)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->bin_count), kNumBins * sizeof(int)));
    for (auto &sl : ctx->slots) {
        for (auto &e : sl.ev) HIPCHK(hipEventCreate(&e));
        for (auto &ph : sl.ev_cls) for (auto &c : ph) for (auto &e : c) HIPCHK(hipEventCreate(&e));
    }
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream_b, hipStreamNonBlocking));
    HIPCHK(hipStreamCreateWithFlags(&ctx->stream_c, hipStreamNonBlocking));
    HIPCHK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    for (auto &t : ctx->ev_tile) for (auto &e : t) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    ctx->cache_budget = prop.totalGlobalMem / 4;
    if (const char *e = getenv("BSPGEMM_FLOW"))
        ctx->flow = !strcmp(e, "fused") ? BSPGEMM_FLOW_FUSED : !strcmp(e, "exact") ? BSPGEMM_FLOW_EXACT : (!strcmp(e, "upper-bound") || !strcmp(e, "ub")) ? BSPGEMM_FLOW_UPPER_BOUND
                                                                                                               : BSPGEMM_FLOW_AUTO;
    if (const char *e = getenv("BSPGEMM_CLASS_TIMING")) ctx->class_timing = atoi(e) != 0;
    if (const char *e = getenv("BSPGEMM_CLASS_STREAMS")) { const int n = atoi(e); ctx->class_streams = n < 1 ? 1 : (n > 3 ? 3 : n); }
    ctx->check = getenv("BSPGEMM_CHECK") != nullptr;
    if (const char *e = getenv("BSPGEMM_RW_BLK")) ctx->rw_blk = atoi(e) ? 1 : 0;
    ctx->debug_alloc = getenv("BSPGEMM_DEBUG_ALLOC") != nullptr;
    ctx->dropin_timing = getenv("BSPGEMM_DROPIN_TIMING") != nullptr;
    if (ctx->debug_alloc)
        fprintf(stderr, "[bspgemm] device %d: %s, %zu MiB, %d CUs; result cache budget %zu MiB\n", device,
                prop.gcnArchName, (size_t)(prop.totalGlobalMem >> 20), prop.multiProcessorCount, ctx->cache_budget >> 20);
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
    hipFree(ctx->rec); hipFree(ctx->recpre); hipFree(ctx->ab); hipFree(ctx->Fmask); hipFree(ctx->hpartials);
    hipFree(ctx->hub_rec); hipFree(ctx->hub_pre);
    hipFree(ctx->tiles); hipFree(ctx->chain); hipFree(ctx->marks8); hipFree(ctx->tile_count); hipFree(ctx->tile_bound);
    hipFree(ctx->pack_totals); hipFree(ctx->tickets);
    if (ctx->h) hipHostFree(ctx->h);
    hipFree(ctx->d_prep);
    hipFree(ctx->chunk_row);
    for (auto &sl : ctx->slots) {
        for (auto &e : sl.ev) if (e) hipEventDestroy(e);
        for (auto &ph : sl.ev_cls) for (auto &c : ph) for (auto &e : c) if (e) hipEventDestroy(e);
    }
    for (auto &c : ctx->cache) if (c.p) hipFree(c.p);
    if (ctx->stream_b) { hipStreamSynchronize(ctx->stream_b); hipStreamDestroy(ctx->stream_b); }
    if (ctx->stream_c) { hipStreamSynchronize(ctx->stream_c); hipStreamDestroy(ctx->stream_c); }
    if (ctx->ev_join) hipEventDestroy(ctx->ev_join);
    hipFree(ctx->stitch_partials);
    for (auto &t : ctx->ev_tile) for (auto &e : t) if (e) hipEventDestroy(e);
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
    auto bail = [&](bspgemm_status st) { bspgemm_matrix_free(m); return st; };   // handle + device arrays
    // +1 int of slack on col_idx so an empty matrix still has a valid pointer
    HIPCHK_B(hipMalloc(reinterpret_cast<void **>(&m->d_row_ptr), ((size_t)rows + 1) * sizeof(int)));
    HIPCHK_B(hipMalloc(reinterpret_cast<void **>(&m->d_col_idx), ((size_t)nnz + 1) * sizeof(int)));
    HIPCHK_B(hipMemcpyAsync(m->d_row_ptr, row_ptr, ((size_t)rows + 1) * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    if (nnz > 0)
        HIPCHK_B(hipMemcpyAsync(m->d_col_idx, col_idx + base, (size_t)nnz * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    launch_rebase_i32(m->d_row_ptr, rows + 1, (int)base, ctx->stream);
    if (bspgemm_status st = ensure_deg8(m)) return bail(st);
    HIPCHK_B(hipStreamSynchronize(ctx->stream));
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
    hipSetDevice(m->ctx->device);
    if (m->owned) {
        hipFree(m->d_row_ptr);
        hipFree(m->d_col_idx);
    }
    hipFree(m->d_deg8);
    hipFree(m->d_blk8);
    delete m;
}
extern "C" bspgemm_status bspgemm_matrix_invalidate(bspgemm_matrix *m)
{
    if (!m) return FAIL(BSPGEMM_ERR_INVALID, "matrix is NULL");
    if (bspgemm_status st = use_device(m->ctx)) return st;
    HIPCHK(hipStreamSynchronize(m->ctx->stream));          // a multiply may still be reading the tables
    hipFree(m->d_deg8);
    hipFree(m->d_blk8);
    m->d_deg8 = nullptr;
    m->d_blk8 = nullptr;
    m->blk8_state = 0;
    return BSPGEMM_OK;
}

extern "C" const char *bspgemm_build_info(void)
{
    return "libbspgemm: HIP kernels for gfx950 only; flows upper-bound (default), exact, fused; "
           "timing-only ablation switches: none (BSP_ABLATE=0); tuning constants are compile-time";
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
    hipFree(ctx->rec); hipFree(ctx->recpre); hipFree(ctx->Fmask); hipFree(ctx->hpartials);
    ctx->F = ctx->Fprefix = ctx->partials = ctx->recpre = ctx->Fmask = ctx->hpartials = nullptr;
    ctx->cnt = ctx->bin_tiles = nullptr;
    ctx->rec = nullptr;
    ctx->rows_cap = 0;
    const size_t cap = rows + rows / 8 + 64;
    const size_t tiles = cap / 2048 + 2;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->F), cap * sizeof(long long)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->Fmask), cap * sizeof(long long)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->Fprefix), (cap + 1) * sizeof(long long)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->partials), (tiles + 1) * sizeof(long long)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->hpartials), (tiles + 1) * sizeof(long long)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->cnt), cap * sizeof(int)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->bin_tiles), (tiles + 1) * kNumBins * sizeof(int)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->rec), cap * sizeof(RowRec)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->recpre), cap * sizeof(long long)));
    if (!ctx->hub_rec) {
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->hub_rec), kHeavySortMax * sizeof(RowRec)));
        HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->hub_pre), kHeavySortMax * sizeof(long long)));
    }
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

static bspgemm_status ensure_fused(bspgemm_context *ctx, size_t rows)
{
    if (rows <= ctx->fused_cap) return BSPGEMM_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    hipFree(ctx->tiles); hipFree(ctx->chain); hipFree(ctx->marks8); hipFree(ctx->tile_count); hipFree(ctx->tile_bound);
    hipFree(ctx->pack_totals); hipFree(ctx->tickets);
    ctx->tiles = nullptr; ctx->chain = nullptr; ctx->marks8 = nullptr; ctx->tile_count = nullptr; ctx->tile_bound = nullptr;
    ctx->pack_totals = nullptr; ctx->tickets = nullptr;
    ctx->fused_cap = 0;
    const size_t cap = rows + rows / 8 + 64;               // a tile holds at least one row
    const size_t blocks = cap / 2048 + 2;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->tiles), cap * sizeof(TileDesc)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->chain), cap * sizeof(unsigned long long) + 256));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->marks8), blocks * 256));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->tile_count), blocks * sizeof(int)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->tile_bound), blocks * sizeof(long long)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->pack_totals), 2 * sizeof(long long)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->tickets), (8 * 32 + 32) * sizeof(unsigned)));
    ctx->fused_cap = cap;
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
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&ctx->tmp), cap * sizeof(int));
    if (e == hipErrorOutOfMemory) {                     // the cache of freed results may hold what is missing
        for (auto &c : ctx->cache) if (c.p) { hipFree(c.p); c.p = nullptr; }
        (void)hipGetLastError();
        e = hipMalloc(reinterpret_cast<void **>(&ctx->tmp), cap * sizeof(int));
    }
    HIPCHK(e);
    ctx->tmp_cap = cap;
    return BSPGEMM_OK;
}

static bspgemm_status ensure_chunk_rows(bspgemm_context *ctx, size_t entries)
{
    if (entries <= ctx->chunk_cap) return BSPGEMM_OK;
    HIPCHK(hipStreamSynchronize(ctx->stream));
    hipFree(ctx->chunk_row);
    ctx->chunk_row = nullptr;
    ctx->chunk_cap = 0;
    const size_t cap = entries + entries / 16 + 64;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->chunk_row), cap * sizeof(int)));
    ctx->chunk_cap = cap;
    return BSPGEMM_OK;
}

// result buffers: best fit from the context's cache of freed results, else hipMalloc
static int result_cache_find(const bspgemm_context *ctx, size_t bytes)
{
    int best = -1;
    for (int i = 0; i < 8; i++) {
        const auto &c = ctx->cache[i];
        if (c.p && c.bytes >= bytes && c.bytes <= 2 * bytes + (1 << 20) &&
            (best < 0 || c.bytes < ctx->cache[best].bytes))
            best = i;
    }
    return best;
}
static bool result_cached(const bspgemm_context *ctx, size_t bytes) { return result_cache_find(ctx, bytes) >= 0; }

static hipError_t result_alloc(bspgemm_context *ctx, void **out, size_t bytes)
{
    const int best = result_cache_find(ctx, bytes);
    if (best >= 0) {
        *out = ctx->cache[best].p;
        ctx->cache[best].p = nullptr;
        return hipSuccess;
    }
    const bool dbg = ctx->debug_alloc;
    const auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(out, bytes);
    if (dbg) {
        fprintf(stderr, "[bspgemm] hipMalloc(%zu) %.3f ms; cache:", bytes,
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        for (const auto &c : ctx->cache) if (c.p) fprintf(stderr, " %zu", c.bytes);
        fprintf(stderr, "\n");
    }
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
    // the cache is capped by BYTES as well as by slots: freed results of a large product must not
    // pin the memory the next one needs (the smallest buffers go first)
    const bool dbg = ctx->debug_alloc;
    size_t held = 0;
    for (const auto &c : ctx->cache) if (c.p) held += c.bytes;
    if (dbg && held + bytes > ctx->cache_budget)
        fprintf(stderr, "[bspgemm] result cache over budget: holds %zu, released %zu, budget %zu\n", held, bytes, ctx->cache_budget);
    while (held + bytes > ctx->cache_budget && held > 0) {
        int small = -1;
        for (int i = 0; i < 8; i++)
            if (ctx->cache[i].p && (small < 0 || ctx->cache[i].bytes < ctx->cache[small].bytes)) small = i;
        if (small < 0) break;
        held -= ctx->cache[small].bytes;
        hipFree(ctx->cache[small].p);
        ctx->cache[small].p = nullptr;
    }
    if (bytes > ctx->cache_budget) { hipFree(p); return; }
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


// class launch order of a phase: the heavy rows first (few long-running workgroups: started early
// they finish under the other classes instead of being the phase's tail), then the one-wave
// classes by capacity
// Launch order of the classes of one phase: order[1..kNumBins-1], alternating over the class streams.  The heavy
// rows first (few long-running workgroups: started early they finish under the other classes instead of being
// the phase's tail), then the one-wave classes LARGEST WORK FIRST (rows x capacity): a phase then ends with its
// small launches, whose ramp-down is short, instead of with the 24-32-chunk classes (measured on the bench
// matrix against ascending capacity: numeric phase 3.74 -> 3.60 ms, step -1.7 %; descending capacity -0.5 %;
// `profiles/r03_ab_class_order.log`).  Where the heavy classes are a large part of the product the one-wave
// classes queue behind them and ascending capacity measured better (power-law: +0.8 % otherwise): kept there.
static void class_order(const int *bin_count, long long total_products, int *order)
{
    order[0] = 0;
    order[1] = kDenseBin;
    order[2] = kMidBin;
    for (int pos = 3; pos < kNumBins; pos++) order[pos] = pos - 2;
    const long long heavy_lower_bound = ((long long)bin_count[kMidBin] + bin_count[kDenseBin]) * kMaxWaveCap;
    if (heavy_lower_bound * 8 >= total_products) return;
    long long key[kNumBins] = {};
    for (int b = 1; b <= kWaveBins; b++) key[b] = (long long)bin_count[b] * kWaveChunks[b];
    for (int a = 3; a < kNumBins; a++)                            // insertion sort of 16 entries, stable
        for (int c = a; c > 3 && key[order[c]] > key[order[c - 1]]; c--) {
            const int t = order[c];
            order[c] = order[c - 1];
            order[c - 1] = t;
        }
}

// the hub rows (class kDenseBin) of a multiply, largest first when there are few enough to rank
static void hub_order(bspgemm_context *ctx, int b, int n, const RowRec *&rec, const long long *&recpre, hipStream_t sx)
{
    if (b != kDenseBin || n < 2 || n > kHeavySortMax) return;
    launch_order_heavy(rec, recpre, n, ctx->hub_rec, ctx->hub_pre, sx);
    rec = ctx->hub_rec;
    recpre = ctx->hub_pre;
}

// closes the multiply's stat slot (its events have all completed: the caller has synchronised)
static void close_slot(bspgemm_context *ctx, int R, const HostScalars *h, long long products, long long nnz_c,
                       const int (*cls_n)[kNumBins], int mid_cap)
{
    bspgemm_context::StatSlot &sl = ctx->slots[ctx->slot_head];
    sl.R = R;
    sl.cls_timed = ctx->class_timing;
    sl.mid_cap = mid_cap;
    sl.h = *h;
    sl.products = products;
    sl.nnz_c = nnz_c;
    memcpy(sl.cls_n, cls_n, sizeof sl.cls_n);
    sl.used = true;
}

static void fill_stats(const bspgemm_context::StatSlot &sl, bspgemm_stats &st)
{
    memset(&st, 0, sizeof st);
    const int R = sl.R;
    st.rows = R;
    st.nnz_a = R > 0 ? (long long)sl.h.a_hi - sl.h.a_lo : 0;
    st.products = sl.products;
    st.nnz_c = sl.nnz_c;
    st.bytes_alg = 4ll * (R + 1) + 12ll * st.nnz_a + 4ll * st.products + 4ll * sl.nnz_c + 8ll * (R + 1);
    st.bytes_read_alg = st.bytes_alg - 4ll * sl.nnz_c - 8ll * (R + 1);
    static_assert(kMaxBins == BSPGEMM_MAX_BINS && kNumBins <= kMaxBins, "stats arrays hold every class");
    st.bins = kNumBins;
    for (int b = 0; b < kNumBins; b++) {
        st.rows_per_bin[b] = sl.h.bin_count[b];
        st.bin_cap[b] = b == 0 ? 0 : (b == kDenseBin ? 0x7fffffff : (b == kMidBin ? sl.mid_cap : 64 * kWaveChunks[b]));
    }
    hipEventElapsedTime(&st.ms_total, sl.ev[0], sl.ev[4]);
    hipEventElapsedTime(&st.ms_prepass, sl.ev[0], sl.ev[1]);
    hipEventElapsedTime(&st.ms_count, sl.ev[1], sl.ev[2]);
    hipEventElapsedTime(&st.ms_symbolic, sl.ev[0], sl.ev[2]);
    hipEventElapsedTime(&st.ms_numeric, sl.ev[2], sl.ev[3]);
    hipEventElapsedTime(&st.ms_stitch, sl.ev[3], sl.ev[4]);
    for (int ph = 0; ph < 2; ph++)
        for (int b = 1; b < kNumBins; b++)
            if (sl.cls_n[ph][b] > 0 && sl.cls_timed) {
                float ms = 0, t0 = 0;
                hipEventElapsedTime(&ms, sl.ev_cls[ph][b][0], sl.ev_cls[ph][b][1]);
                hipEventElapsedTime(&t0, sl.ev[0], sl.ev_cls[ph][b][0]);
                (ph == 0 ? st.ms_bin_count : st.ms_bin)[b] = ms;
                (ph == 0 ? st.t_bin_count : st.t_bin)[b] = t0;
            }
}

// C rows [row_begin,row_end) of A*B:  symbolic (row work -> classes -> EXACT row sizes -> scan =
// C.row_ptr) then numeric (every one-wave row emitted at its final place in a C.col_idx of exactly
// nnz(C) entries) -- the two passes BASELINE.json's north star names.  Heavy rows (F_i > 2048) are
// accumulated and read out once, during the symbolic phase, into a workspace bounded by
// sum(min(F_i, cols)), and moved to their place during the numeric phase.
static bspgemm_status multiply_exact(bspgemm_context *ctx, const bspgemm_matrix *A, const bspgemm_matrix *B,
                                     int row_begin, int row_end, bspgemm_result **out)
{
    if (!out) return FAIL(BSPGEMM_ERR_INVALID, "result pointer is NULL");
    *out = nullptr;
    if (bspgemm_status st = check_operands(ctx, A, B, row_begin, row_end)) return st;
    if (bspgemm_status st = use_device(ctx)) return st;
    const int R = row_end - row_begin;
    hipStream_t s = ctx->stream, sB = ctx->stream_b, sC = ctx->stream_c;
    if (bspgemm_status st = ensure_rows(ctx, (size_t)R + 1)) return st;
    if (bspgemm_status st = ensure_ab(ctx, (size_t)A->nnz + 1)) return st;

    bspgemm_result *C = new (std::nothrow) bspgemm_result{ctx, R, 0, nullptr, nullptr, 0};
    if (!C) return FAIL(BSPGEMM_ERR_ALLOC, "result");
    auto bail = [&](bspgemm_status st) {                  // nothing of C may still be written when it is released
        hipStreamSynchronize(s); hipStreamSynchronize(sB); hipStreamSynchronize(sC);
        bspgemm_result_free(C);
        return st;
    };
    ctx->slot_head = (ctx->slot_head + 1) % bspgemm_context::kStatSlots;
    bspgemm_context::StatSlot &slot = ctx->slots[ctx->slot_head];
    slot.used = false;

    HIPCHK_B(hipEventRecord(slot.ev[0], s));
    HIPCHK_B(result_alloc(ctx, reinterpret_cast<void **>(&C->d_row_ptr), result_bytes_rowptr(R)));

    // ---- symbolic 1: per-row products, their prefix, capacity classes ---------------------
    const int scan_tiles = (R + 2047) / 2048;
    const int heavy_cols = B->cols > 0 ? B->cols : 1;
    // the extents ab[] come from the prepass, as in the other flow: both passes of the one-wave classes read them
    if (bspgemm_status st = ensure_blk8(B)) return bail(st);       // (wrapped device arrays: first use)
    launch_row_work(A->d_row_ptr, A->d_col_idx, B->d_row_ptr, B->blk8_state == 1 ? B->d_blk8 : nullptr, row_begin, row_end,
                    ctx->F, ctx->ab, s);
    launch_scan_and_bin(ctx->F, R, row_begin, A->d_row_ptr, ctx->Fprefix, ctx->partials, ctx->bin_tiles,
                        ctx->bin_count, ctx->rec, ctx->recpre, ctx->cnt, heavy_cols, ctx->hpartials, mid_cap_for_cols(B->cols), 0, s);
    HostScalars *h = ctx->h;
    HIPCHK_B(hipMemcpyAsync(&h->totalF, ctx->Fprefix + R, sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipMemcpyAsync(&h->heavy_total, ctx->hpartials + scan_tiles, sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipMemcpyAsync(h->bin_count, ctx->bin_count, kNumBins * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipMemcpyAsync(&h->a_lo, A->d_row_ptr + row_begin, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipMemcpyAsync(&h->a_hi, A->d_row_ptr + row_end, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipEventRecord(slot.ev[1], s));
    HIPCHK_B(hipStreamSynchronize(s));
    const long long totalF = R > 0 ? h->totalF : 0;
    if (R == 0) { memset(h->bin_count, 0, sizeof h->bin_count); h->heavy_total = 0; }
    if (bspgemm_status st = ensure_tmp(ctx, (size_t)h->heavy_total + 1)) return bail(st);

    size_t bin_start[kNumBins + 1] = {0, 0};               // class b's segment of rec[] (class 0 has none)
    for (int b = 1; b < kNumBins; b++) bin_start[b + 1] = bin_start[b] + (size_t)h->bin_count[b];
    int cls_n[2][kNumBins] = {};
    hipStream_t lanes[3] = {s, sB, sC};
    const int nlanes = ctx->class_streams;
    // The class launches of a phase are independent (disjoint rows): they alternate over two streams
    // so that one launch's draining tail overlaps the next one's ramp-up.  The side streams start
    // behind the phase's inputs (fork) and the main stream waits for them at its end (join).
    auto fork = [&](hipEvent_t ev) -> hipError_t {
        if (hipError_t e = hipEventRecord(ev, s)) return e;
        for (int l = 1; l < 3; l++)
            if (hipError_t e = hipStreamWaitEvent(lanes[l], ev, 0)) return e;
        return hipSuccess;
    };
    auto join = [&](int l, hipEvent_t ev) -> hipError_t {
        if (hipError_t e = hipEventRecord(ev, lanes[l])) return e;
        return hipStreamWaitEvent(s, ev, 0);
    };

    // ---- symbolic 2: exact |C_i| of every row, scanned into C.row_ptr -----------------------
    int order[kNumBins];
    if (R > 0) {
        class_order(h->bin_count, totalF, order);
        HIPCHK_B(fork(ctx->ev_tile[0][0]));
        for (int pos = 1; pos < kNumBins; pos++) {
            const int b = order[pos];
            const int n = h->bin_count[b];
            cls_n[0][b] = n;
            if (n <= 0) continue;
            hipStream_t sx = lanes[pos % nlanes];
            const RowRec *rec = ctx->rec + bin_start[b];
            if (ctx->class_timing) HIPCHK_B(hipEventRecord(slot.ev_cls[0][b][0], sx));
            if (b <= kWaveBins) {
                // the numeric kernel without its emit half: |C_i| = F_i as soon as every product is seen to sit alone
                // in its 32-column slot, the level-0 masks are only built and counted for the other rows
                launch_wave_rows(b, wave_levels_for_cols(B->cols), ctx->ab, B->d_col_idx, B->cols, rec, nullptr, nullptr, n,
                                 row_begin, nullptr, ctx->cnt, sx, true);
            } else {
                const long long *hpre = ctx->recpre + bin_start[b];
                hub_order(ctx, b, n, rec, hpre, sx);
                HIPCHK_B(launch_dense_rows(b == kMidBin, ctx->ab, B->d_col_idx, B->cols, rec, hpre, n,
                                           row_begin, ctx->tmp, ctx->cnt, sx));
            }
            if (ctx->class_timing) HIPCHK_B(hipEventRecord(slot.ev_cls[0][b][1], sx));
        }
        HIPCHK_B(hipGetLastError());
        for (int l = 1; l < nlanes; l++) HIPCHK_B(join(l, ctx->ev_tile[0][l]));
        launch_scan_counts(ctx->cnt, R, C->d_row_ptr, ctx->partials, nullptr, s);
    } else {
        HIPCHK_B(hipMemsetAsync(C->d_row_ptr, 0, sizeof(long long), s));
    }
    HIPCHK_B(hipMemcpyAsync(&h->nnzC, C->d_row_ptr + R, sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipEventRecord(slot.ev[2], s));

    // C.col_idx: nnz(C) <= F entries are needed.  A cached buffer that holds F entries is taken
    // without waiting for nnz(C); otherwise the size is read back and exactly that is allocated
    // (F itself when it is within 2 % of nnz(C): the next product of this shape then finds it cached).
    bool synced = false;
    const bool check = ctx->check;                         // development: never emit on unverified sizes
    if (!check && result_cached(ctx, result_bytes_colidx(totalF))) {
        HIPCHK_B(result_alloc(ctx, reinterpret_cast<void **>(&C->d_col_idx), result_bytes_colidx(totalF)));
        C->col_cap = totalF;
    } else {
        HIPCHK_B(hipStreamSynchronize(s));
        synced = true;
        if (h->nnzC < 0 || h->nnzC > totalF) return bail(FAIL(BSPGEMM_ERR_HIP, "symbolic pass counted more outputs than products"));
        const long long want = (totalF - h->nnzC <= h->nnzC / 50 + 4096) ? totalF : h->nnzC;
        HIPCHK_B(result_alloc(ctx, reinterpret_cast<void **>(&C->d_col_idx), result_bytes_colidx(want)));
        C->col_cap = want;
    }

    // ---- numeric: every row emitted at its final place ------------------------------------------
    if (R > 0) {
        const int levels = wave_levels_for_cols(B->cols);
        HIPCHK_B(fork(ctx->ev_tile[1][0]));
        for (int pos = 1; pos < kNumBins; pos++) {
            const int b = order[pos];
            const int n = h->bin_count[b];
            cls_n[1][b] = n;
            if (n <= 0) continue;
            const RowRec *rec = ctx->rec + bin_start[b];
            const long long *recpre = ctx->recpre + bin_start[b];
            // the heavy rows' move runs beside the class launches on the third stream
            hipStream_t sx = b > kWaveBins ? sC : lanes[pos % nlanes];
            if (ctx->class_timing) HIPCHK_B(hipEventRecord(slot.ev_cls[1][b][0], sx));
            if (b <= kWaveBins)
                launch_wave_rows(b, levels, ctx->ab, B->d_col_idx, B->cols, rec, recpre, C->d_row_ptr, n, row_begin,
                                 C->d_col_idx, nullptr, sx);
            else
                launch_place_heavy(ctx->tmp, rec, recpre, n, C->d_row_ptr, row_begin, C->d_col_idx, sx);
            if (ctx->class_timing) HIPCHK_B(hipEventRecord(slot.ev_cls[1][b][1], sx));
        }
        HIPCHK_B(hipGetLastError());
        for (int l = 1; l < nlanes; l++) HIPCHK_B(join(l, ctx->ev_tile[1][l]));
    }
    HIPCHK_B(hipEventRecord(slot.ev[3], s));
    if (R > 0) HIPCHK_B(join(2, ctx->ev_join));            // the heavy rows' move (third stream)
    HIPCHK_B(hipEventRecord(slot.ev[4], s));
    HIPCHK_B(hipStreamSynchronize(s));
    (void)synced;
    C->nnz = h->nnzC;
    close_slot(ctx, R, h, totalF, C->nnz, cls_n, mid_cap_for_cols(B->cols));
    *out = C;
    return BSPGEMM_OK;
}

// C rows [row_begin,row_end) of A*B in ONE pass over the products, rows written once at their final
// place (csrc/tile_rows.inc): products per row -> tiles of consecutive rows -> one persistent kernel
// that accumulates tile by tile in row order and chains the tiles' sizes by look-back -- the
// reference's own order of events (final/SpGEMM_mpi_omp.c:28-42: a row is appended where the
// previous one ended).  Heavy rows (F_i > 2048) are accumulated first, into a workspace bounded by
// sum(min(F_i, cols)); the chain carries their sizes and they are moved once C.row_ptr exists.
static bspgemm_status multiply_fused(bspgemm_context *ctx, const bspgemm_matrix *A, const bspgemm_matrix *B,
                                     int row_begin, int row_end, bspgemm_result **out)
{
    if (!out) return FAIL(BSPGEMM_ERR_INVALID, "result pointer is NULL");
    *out = nullptr;
    if (bspgemm_status st = check_operands(ctx, A, B, row_begin, row_end)) return st;
    if (bspgemm_status st = use_device(ctx)) return st;
    const int R = row_end - row_begin;
    hipStream_t s = ctx->stream, sB = ctx->stream_b, sC = ctx->stream_c;
    if (bspgemm_status st = ensure_rows(ctx, (size_t)R + 1)) return st;
    if (bspgemm_status st = ensure_fused(ctx, (size_t)R + 1)) return st;
    if (bspgemm_status st = ensure_ab(ctx, (size_t)A->nnz + 1)) return st;    // (heavy rows' extents)

    bspgemm_result *C = new (std::nothrow) bspgemm_result{ctx, R, 0, nullptr, nullptr, 0};
    if (!C) return FAIL(BSPGEMM_ERR_ALLOC, "result");
    auto bail = [&](bspgemm_status st) {
        hipStreamSynchronize(s); hipStreamSynchronize(sB); hipStreamSynchronize(sC);
        bspgemm_result_free(C);
        return st;
    };
    ctx->slot_head = (ctx->slot_head + 1) % bspgemm_context::kStatSlots;
    bspgemm_context::StatSlot &slot = ctx->slots[ctx->slot_head];
    slot.used = false;

    HIPCHK_B(hipEventRecord(slot.ev[0], s));
    HIPCHK_B(result_alloc(ctx, reinterpret_cast<void **>(&C->d_row_ptr), result_bytes_rowptr(R)));

    // ---- products per row, their prefix, the heavy rows' records, the tiles ------------------------
    const int scan_tiles = (R + 2047) / 2048;
    const int heavy_cols = B->cols > 0 ? B->cols : 1;
    if (bspgemm_status st = ensure_deg8(B)) return bail(st);
    if (bspgemm_status st = ensure_blk8(B)) return bail(st);
    launch_row_products(A->d_row_ptr, A->d_col_idx, B->d_row_ptr, B->d_deg8, row_begin, row_end, ctx->F, s);
    launch_scan_and_bin(ctx->F, R, row_begin, A->d_row_ptr, ctx->Fprefix, ctx->partials, ctx->bin_tiles,
                        ctx->bin_count, ctx->rec, ctx->recpre, ctx->cnt, heavy_cols, ctx->hpartials, mid_cap_for_cols(B->cols), 0, s);
    // rows a tile is expected to hold, from the operands' mean row lengths (decides the key width)
    const double mean_f = (A->rows > 0 && B->rows > 0) ? ((double)A->nnz / A->rows) * ((double)B->nnz / B->rows) : 0.0;
    int row_bits = 0, col_bits = 0;
    const int tile_cap = kTileCap;
    const int levels = tile_levels_for(B->cols, mean_f >= 1.0 ? (long long)(tile_cap / mean_f + 0.999) : 64, tile_cap, &row_bits, &col_bits);
    const int maxr = (1 << row_bits) > 63 ? 63 : (1 << row_bits);
    launch_pack_tiles_count(ctx->F, R, tile_cap, maxr, heavy_cols, ctx->marks8, ctx->tile_count, ctx->tile_bound,
                            ctx->pack_totals, s);
    launch_pack_tiles_emit(ctx->F, R, tile_cap, maxr, A->d_row_ptr + row_begin, ctx->marks8, ctx->tile_count, ctx->tiles, s);
    HostScalars *h = ctx->h;
    HIPCHK_B(hipMemcpyAsync(&h->totalF, ctx->Fprefix + R, sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipMemcpyAsync(&h->heavy_total, ctx->hpartials + scan_tiles, sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipMemcpyAsync(h->pack_totals, ctx->pack_totals, 2 * sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipMemcpyAsync(h->bin_count, ctx->bin_count, kNumBins * sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipMemcpyAsync(&h->a_lo, A->d_row_ptr + row_begin, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipMemcpyAsync(&h->a_hi, A->d_row_ptr + row_end, sizeof(int), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipEventRecord(slot.ev[1], s));
    HIPCHK_B(hipStreamSynchronize(s));
    const long long totalF = R > 0 ? h->totalF : 0;
    if (R == 0) { memset(h->bin_count, 0, sizeof h->bin_count); h->heavy_total = 0; h->pack_totals[0] = h->pack_totals[1] = 0; }
    const long long ntiles_ll = h->pack_totals[0], bound = h->pack_totals[1];
    if (ntiles_ll < 0 || ntiles_ll > R) return bail(FAIL(BSPGEMM_ERR_HIP, "tile packer returned an impossible tile count"));
    const int ntiles = (int)ntiles_ll;
    if (bspgemm_status st = ensure_tmp(ctx, (size_t)h->heavy_total + 1)) return bail(st);
    HIPCHK_B(result_alloc(ctx, reinterpret_cast<void **>(&C->d_col_idx), result_bytes_colidx(bound)));
    C->col_cap = bound;

    size_t bin_start[kNumBins + 1] = {0, 0};
    for (int b = 1; b < kNumBins; b++) bin_start[b + 1] = bin_start[b] + (size_t)h->bin_count[b];
    int cls_n[2][kNumBins] = {};
    // ---- heavy rows: accumulated and read out into the workspace, sizes to cnt[] -------------------
    const int heavy_bins[2] = {kDenseBin, kMidBin};
    hipStream_t heavy_lane[2] = {sB, sC};
    bool any_heavy = false;
    for (int k = 0; k < 2; k++) any_heavy = any_heavy || h->bin_count[heavy_bins[k]] > 0;
    if (any_heavy) {
        HIPCHK_B(hipEventRecord(ctx->ev_tile[0][0], s));
        for (int k = 0; k < 2; k++) {
            const int b = heavy_bins[k], n = h->bin_count[b];
            cls_n[0][b] = n;
            if (n <= 0) continue;
            hipStream_t sx = heavy_lane[k];
            HIPCHK_B(hipStreamWaitEvent(sx, ctx->ev_tile[0][0], 0));
            const RowRec *rec = ctx->rec + bin_start[b];
            const long long *hpre = ctx->recpre + bin_start[b];
            if (ctx->class_timing) HIPCHK_B(hipEventRecord(slot.ev_cls[0][b][0], sx));
            hub_order(ctx, b, n, rec, hpre, sx);
            launch_extents_of_rows(rec, n, A->d_col_idx, B->d_row_ptr, ctx->ab, sx);
            HIPCHK_B(launch_dense_rows(b == kMidBin, ctx->ab, B->d_col_idx, B->cols, rec, hpre, n,
                                       row_begin, ctx->tmp, ctx->cnt, sx));
            if (ctx->class_timing) HIPCHK_B(hipEventRecord(slot.ev_cls[0][b][1], sx));
            HIPCHK_B(hipEventRecord(ctx->ev_tile[0][1 + k], sx));
            HIPCHK_B(hipStreamWaitEvent(s, ctx->ev_tile[0][1 + k], 0));
        }
    }
    HIPCHK_B(hipEventRecord(slot.ev[2], s));

    // ---- the tiles ---------------------------------------------------------------------------------
    if (ntiles > 0) {
        // chain words: 4 B per tile, then 2 x 8 B per block of 64 tiles
        const size_t nblk = ((size_t)ntiles + 63) / 64;
        const size_t bacc_off = (((size_t)ntiles * sizeof(unsigned) + 15) & ~(size_t)15);
        const size_t chain_bytes = bacc_off + 2 * nblk * sizeof(unsigned long long);
        HIPCHK_B(hipMemsetAsync(ctx->chain, 0, chain_bytes, s));
        HIPCHK_B(hipMemsetAsync(ctx->tickets, 0, (8 * 32 + 32) * sizeof(unsigned), s));
        int grid = tile_rows_grid(levels, ctx->device);
        if (grid < 1) grid = 1;
        if (grid > ntiles) grid = ntiles;
        TileArgs ta;
        ta.Arow = A->d_row_ptr + row_begin;
        ta.Acol = A->d_col_idx;
        ta.Brow = B->d_row_ptr;
        ta.Bblk = B->blk8_state == 1 ? B->d_blk8 : nullptr;
        ta.Bcol = B->d_col_idx;
        ta.tiles = ctx->tiles;
        ta.ntiles = ntiles;
        ta.nrows = R;
        ta.cnt = ctx->cnt;
        ta.tdesc = reinterpret_cast<unsigned *>(ctx->chain);
        ta.bacc = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(ctx->chain) + bacc_off);
        ta.binc = ta.bacc + nblk;
        ta.ticket = ctx->tickets;
        ta.nshards = grid >= 256 ? 8 : 1;
        ta.err = ctx->tickets + 8 * 32;
        ta.row_ptr = C->d_row_ptr;
        ta.col_idx = C->d_col_idx;
        ta.col_bits = col_bits;
        cls_n[1][1] = ntiles;
        if (ctx->class_timing) HIPCHK_B(hipEventRecord(slot.ev_cls[1][1][0], s));
        HIPCHK_B(launch_tile_rows(levels, ta, grid, s));
        if (ctx->class_timing) HIPCHK_B(hipEventRecord(slot.ev_cls[1][1][1], s));
        HIPCHK_B(hipMemcpyAsync(&h->chain_err, ta.err, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    } else {
        HIPCHK_B(hipMemsetAsync(C->d_row_ptr, 0, sizeof(long long), s));
        h->chain_err = 0;
    }
    HIPCHK_B(hipEventRecord(slot.ev[3], s));
    for (int k = 0; k < 2; k++) {
        const int b = heavy_bins[k], n = h->bin_count[b];
        if (n > 0) launch_place_heavy(ctx->tmp, ctx->rec + bin_start[b], ctx->recpre + bin_start[b], n, C->d_row_ptr, row_begin,
                                      C->d_col_idx, s);
    }
    HIPCHK_B(hipGetLastError());
    HIPCHK_B(hipMemcpyAsync(&h->nnzC, C->d_row_ptr + R, sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipEventRecord(slot.ev[4], s));
    HIPCHK_B(hipStreamSynchronize(s));
    if (h->chain_err) return bail(FAIL(BSPGEMM_ERR_HIP, "fused flow: a tile waited more than 4 s for the tiles before it"));
    if (h->nnzC < 0 || h->nnzC > bound) return bail(FAIL(BSPGEMM_ERR_HIP, "fused flow: more outputs than the bound"));
    C->nnz = h->nnzC;
    close_slot(ctx, R, h, totalF, C->nnz, cls_n, mid_cap_for_cols(B->cols));
    *out = C;
    return BSPGEMM_OK;
}

// C = F .* (A*B).  The mask bounds a row (|C_i| <= |F_i|), usually far below its product count, so
// rows are binned and placed by MASK length in an upper-bound workspace and squeezed together by
// the compaction kernel once the counts are scanned.
static bspgemm_status multiply_upper_bound(bspgemm_context *ctx, const bspgemm_matrix *A,
                                           const bspgemm_matrix *B, const bspgemm_matrix *Fm,
                                           int row_begin, int row_end, bspgemm_result **out)
{
    if (!out) return FAIL(BSPGEMM_ERR_INVALID, "result pointer is NULL");
    *out = nullptr;
    if (bspgemm_status st = check_operands(ctx, A, B, row_begin, row_end)) return st;
    if (Fm && (Fm->ctx != ctx || Fm->rows < row_end)) return FAIL(BSPGEMM_ERR_INVALID, "mask has fewer rows than A / wrong context");
    if (bspgemm_status st = use_device(ctx)) return st;
    const int R = row_end - row_begin;
    hipStream_t s = ctx->stream, sB = ctx->stream_b, sC = ctx->stream_c;
    if (bspgemm_status st = ensure_rows(ctx, (size_t)R + 1)) return st;
    if (bspgemm_status st = ensure_ab(ctx, (size_t)A->nnz + 1)) return st;

    bspgemm_result *C = new (std::nothrow) bspgemm_result{ctx, R, 0, nullptr, nullptr, 0};
    if (!C) return FAIL(BSPGEMM_ERR_ALLOC, "result");
    auto bail = [&](bspgemm_status st) {                  // nothing of C may still be written when it is released
        hipStreamSynchronize(s); hipStreamSynchronize(sB); hipStreamSynchronize(sC);
        bspgemm_result_free(C);
        return st;
    };
    ctx->slot_head = (ctx->slot_head + 1) % bspgemm_context::kStatSlots;
    bspgemm_context::StatSlot &slot = ctx->slots[ctx->slot_head];
    slot.used = false;

    HIPCHK_B(hipEventRecord(slot.ev[0], s));
    HIPCHK_B(result_alloc(ctx, reinterpret_cast<void **>(&C->d_row_ptr), result_bytes_rowptr(R)));
    if (bspgemm_status st = ensure_blk8(B)) return bail(st);
    launch_row_work(A->d_row_ptr, A->d_col_idx, B->d_row_ptr, B->blk8_state == 1 ? B->d_blk8 : nullptr, row_begin, row_end,
                    ctx->F, ctx->ab, s);
    HostScalars *h = ctx->h;
    h->products = 0;
    // rows are classified by their products and placed by min(products, B.cols) -- or, masked, both by
    // the mask row's length (|C_i| <= |F_i|); the true product count is summed separately
    const long long *size_by = ctx->F;
    if (Fm) {
        launch_mask_lengths(ctx->F, Fm->d_row_ptr, row_begin, R, ctx->Fmask, s);
        size_by = ctx->Fmask;
    }
    launch_scan_and_bin(size_by, R, row_begin, A->d_row_ptr, ctx->Fprefix, ctx->partials, ctx->bin_tiles,
                        ctx->bin_count, ctx->rec, ctx->recpre, ctx->cnt, 0, ctx->hpartials, mid_cap_for_cols(B->cols),
                        B->cols > 0 ? B->cols : 1, s, ctx->d_prep, Fm ? ctx->F : nullptr);
    HIPCHK_B(hipMemcpyAsync(&h->prep, ctx->d_prep, sizeof(PrepScalars), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipEventRecord(slot.ev[1], s));
    HIPCHK_B(hipStreamSynchronize(s));
    h->totalF = h->prep.totalF;
    h->products = h->prep.products;
    h->a_lo = h->prep.a_lo;
    h->a_hi = h->prep.a_hi;
    memcpy(h->bin_count, h->prep.bin_count, sizeof h->bin_count);
    const long long total = R > 0 ? h->totalF : 0;         // sum of min(products, cols) (masked: of mask-row lengths): bounds nnz(C)
    if (R == 0) memset(h->bin_count, 0, sizeof h->bin_count);
    if (bspgemm_status st = ensure_tmp(ctx, (size_t)total + 1)) return bail(st);
    if (bspgemm_status st = ensure_chunk_rows(ctx, compact_chunk_rows(total))) return bail(st);
    // C.col_idx: a cached buffer of the upper-bound size is taken now (nothing to wait for); else it
    // is allocated with exactly nnz(C) entries once the counts are scanned
    if (result_cached(ctx, result_bytes_colidx(total))) {
        HIPCHK_B(result_alloc(ctx, reinterpret_cast<void **>(&C->d_col_idx), result_bytes_colidx(total)));
        C->col_cap = total;
    }
    const int levels = wave_levels_for_cols(B->cols);
    HIPCHK_B(hipEventRecord(slot.ev[2], s));               // no count phase here: ev[1]..ev[2] is the host's turn-around

    size_t bin_start[kNumBins + 1] = {0, 0};
    for (int b = 1; b < kNumBins; b++) bin_start[b + 1] = bin_start[b] + (size_t)h->bin_count[b];
    int cls_n[2][kNumBins] = {};
    hipStream_t lanes[3] = {s, sB, sC};
    const int nlanes = ctx->class_streams;
    if (R > 0) {
        HIPCHK_B(hipEventRecord(ctx->ev_tile[0][0], s));
        for (int l = 1; l < nlanes; l++) HIPCHK_B(hipStreamWaitEvent(lanes[l], ctx->ev_tile[0][0], 0));
        int order[kNumBins];
        class_order(h->bin_count, h->products, order);
        for (int pos = 1; pos < kNumBins; pos++) {
            const int b = order[pos];
            const int n = h->bin_count[b];
            cls_n[1][b] = n;
            if (n <= 0) continue;
            hipStream_t sx = lanes[pos % nlanes];
            const RowRec *rec = ctx->rec + bin_start[b];
            const long long *recpre = ctx->recpre + bin_start[b];
            if (ctx->class_timing) HIPCHK_B(hipEventRecord(slot.ev_cls[1][b][0], sx));
            if (!Fm) hub_order(ctx, b, n, rec, recpre, sx);
            if (!Fm && b <= kWaveBins)
                launch_wave_rows(b, levels, ctx->ab, B->d_col_idx, B->cols, rec, recpre, nullptr, n, row_begin,
                                 ctx->tmp, ctx->cnt, sx);
            else if (!Fm)
                HIPCHK_B(launch_dense_rows(b == kMidBin, ctx->ab, B->d_col_idx, B->cols, rec, recpre, n, row_begin, ctx->tmp,
                                           ctx->cnt, sx));
            else if (b <= kWaveBins && wave_masked_supported(B->cols))
                launch_wave_masked(b, ctx->ab, B->d_col_idx, B->cols, Fm->d_row_ptr, Fm->d_col_idx, rec, recpre, n,
                                   row_begin, ctx->tmp, ctx->cnt, sx);
            else
                HIPCHK_B(launch_dense_rows_masked(ctx->ab, B->d_col_idx, B->cols, rec, recpre, n, row_begin,
                                                  ctx->tmp, ctx->cnt, Fm->d_row_ptr, Fm->d_col_idx, sx));
            if (ctx->class_timing) HIPCHK_B(hipEventRecord(slot.ev_cls[1][b][1], sx));
        }
        HIPCHK_B(hipGetLastError());
        for (int l = 1; l < nlanes; l++) {
            HIPCHK_B(hipEventRecord(ctx->ev_tile[0][l], lanes[l]));
            HIPCHK_B(hipStreamWaitEvent(s, ctx->ev_tile[0][l], 0));
        }
        HIPCHK_B(hipEventRecord(slot.ev[3], s));
        launch_scan_counts(ctx->cnt, R, C->d_row_ptr, ctx->partials, nullptr, s, ctx->chunk_row);
    } else {
        HIPCHK_B(hipMemsetAsync(C->d_row_ptr, 0, sizeof(long long), s));
        HIPCHK_B(hipEventRecord(slot.ev[3], s));
    }
    HIPCHK_B(hipMemcpyAsync(&h->nnzC, C->d_row_ptr + R, sizeof(long long), hipMemcpyDeviceToHost, s));
    if (!C->d_col_idx) {
        HIPCHK_B(hipStreamSynchronize(s));
        const long long want = (total - h->nnzC <= h->nnzC / 50 + 4096) ? total : h->nnzC;
        HIPCHK_B(result_alloc(ctx, reinterpret_cast<void **>(&C->d_col_idx), result_bytes_colidx(want)));
        C->col_cap = want;
    }
    if (R > 0) {
        launch_compact(ctx->tmp, ctx->Fprefix, C->d_row_ptr, 0, R, total, C->d_col_idx, s, ctx->chunk_row);
        HIPCHK_B(hipGetLastError());
    }
    HIPCHK_B(hipEventRecord(slot.ev[4], s));
    HIPCHK_B(hipStreamSynchronize(s));
    C->nnz = h->nnzC;
    close_slot(ctx, R, h, R > 0 ? h->products : 0, C->nnz, cls_n, mid_cap_for_cols(B->cols));
    *out = C;
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_multiply(bspgemm_context *ctx, const bspgemm_matrix *A,
                                           const bspgemm_matrix *B, int row_begin, int row_end,
                                           bspgemm_result **out)
{
    // Two ways to the same CSR (tuning): "exact" sizes every row first (symbolic
    // count pass) and emits at the final place: C.col_idx is nnz(C) entries and there is no workspace
    // of F entries; "upper-bound" places rows by their product count and squeezes them together
    // afterwards: faster where duplicates are rare (the compaction streams at HBM rate, the count
    // pass costs most of a numeric pass), but it holds 2F entries.  Default: upper-bound, and exact
    // when that does not fit.
    if (!ctx) return FAIL(BSPGEMM_ERR_INVALID, "ctx is NULL");
    if (ctx->flow == BSPGEMM_FLOW_EXACT) return multiply_exact(ctx, A, B, row_begin, row_end, out);
    if (ctx->flow == BSPGEMM_FLOW_FUSED) return multiply_fused(ctx, A, B, row_begin, row_end, out);
    bspgemm_status st = multiply_upper_bound(ctx, A, B, nullptr, row_begin, row_end, out);
    if (st == BSPGEMM_ERR_ALLOC && ctx->flow == BSPGEMM_FLOW_AUTO) {
        (void)hipGetLastError();
        st = multiply_exact(ctx, A, B, row_begin, row_end, out);
    }
    return st;
}

extern "C" bspgemm_status bspgemm_set_class_timing(bspgemm_context *ctx, int on)
{
    if (!ctx) return FAIL(BSPGEMM_ERR_INVALID, "set_class_timing");
    ctx->class_timing = on != 0;
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_set_flow(bspgemm_context *ctx, int flow)
{
    if (!ctx || flow < BSPGEMM_FLOW_AUTO || flow > BSPGEMM_FLOW_FUSED) return FAIL(BSPGEMM_ERR_INVALID, "set_flow");
    ctx->flow = flow;
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_multiply_masked(bspgemm_context *ctx, const bspgemm_matrix *A,
                                                  const bspgemm_matrix *B, const bspgemm_matrix *F,
                                                  int row_begin, int row_end, bspgemm_result **out)
{
    if (!F) {
        if (out) *out = nullptr;
        return FAIL(BSPGEMM_ERR_INVALID, "mask is NULL");
    }
    return multiply_upper_bound(ctx, A, B, F, row_begin, row_end, out);
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

extern "C" bspgemm_status bspgemm_stats_at(const bspgemm_context *ctx, int age, bspgemm_stats *out)
{
    if (!ctx || !out || age < 0 || age >= bspgemm_context::kStatSlots) return FAIL(BSPGEMM_ERR_INVALID, "stats_at");
    const int k = (ctx->slot_head - age % bspgemm_context::kStatSlots + bspgemm_context::kStatSlots) % bspgemm_context::kStatSlots;
    if (!ctx->slots[k].used) return FAIL(BSPGEMM_ERR_INVALID, "no multiply of that age has completed on this context");
    fill_stats(ctx->slots[k], *out);
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_last_stats(const bspgemm_context *ctx, bspgemm_stats *out)
{
    return bspgemm_stats_at(ctx, 0, out);
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
    auto bail = [&](bspgemm_status st) { bspgemm_matrix_free(m); return st; };
    HIPCHK_B(hipMalloc(reinterpret_cast<void **>(&m->d_row_ptr), ((size_t)C->rows + 1) * sizeof(int)));
    HIPCHK_B(hipMalloc(reinterpret_cast<void **>(&m->d_col_idx), ((size_t)C->nnz + 1) * sizeof(int)));
    launch_narrow_row_ptr(C->d_row_ptr, m->d_row_ptr, C->rows + 1, ctx->stream);
    if (C->nnz > 0)
        HIPCHK_B(hipMemcpyAsync(m->d_col_idx, C->d_col_idx, (size_t)C->nnz * sizeof(int), hipMemcpyDeviceToDevice, ctx->stream));
    if (bspgemm_status st = ensure_deg8(m)) return bail(st);
    HIPCHK_B(hipStreamSynchronize(ctx->stream));
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
    {
        auto bail = [&](bspgemm_status st) { bspgemm_matrix_free(cur); return st; };
        HIPCHK_B(hipMalloc(reinterpret_cast<void **>(&cur->d_row_ptr), ((size_t)n + 1) * sizeof(int)));
        HIPCHK_B(hipMalloc(reinterpret_cast<void **>(&cur->d_col_idx), ((size_t)cur->nnz + 1) * sizeof(int)));
        launch_add_diagonal(A->d_row_ptr, A->d_col_idx, n, cur->d_row_ptr, cur->d_col_idx, ctx->stream);
        if (bspgemm_status st = ensure_deg8(cur)) return bail(st);
        HIPCHK_B(hipStreamSynchronize(ctx->stream));
    }
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
    if (bspgemm_status st = ensure_deg8(B)) return st;
    launch_row_products(A->d_row_ptr, A->d_col_idx, B->d_row_ptr, B->d_deg8, 0, R, ctx->F, ctx->stream);
    launch_scan_and_bin(ctx->F, R, 0, A->d_row_ptr, ctx->Fprefix, ctx->partials, ctx->bin_tiles, ctx->bin_count,
                        ctx->rec, ctx->recpre, ctx->cnt, 0, nullptr, mid_cap_for_cols(B->cols), 0, ctx->stream);
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
    const bool timing = ctx->dropin_timing;                // stage times to stderr
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    // B's row count is implicit in the reference (never passed): 1 + the largest column of A used
    const int brows = bspgemm_par_max_plus_one(Acol + Arow[r0], (long long)Arow[r1] - Arow[r0]);
    // mode 0 (the call the reference's driver makes, :322): the malloc'ed result is sized by an upper bound from
    // one host pass over A, and faulted in and PINNED IN PLACE (hipHostRegister) by a helper thread while the
    // operands are uploaded and multiplied -- the download is then one DMA at the link's rate into the caller's
    // own memory.  (It used to be a pageable copy started after the multiply: 172 of 200 ms on BASELINE
    // config 3, the 5.3 GB result moving at 31 GB/s.)
    long long bound = -1;
    if (mode == 0 && !Frow) bound = bspgemm_par_output_bound(Acol, Arow, r0, r1, Brow, brows, Bm > 0 ? Bm : 1);
    // The helper works through the destination in pieces of 128 MB -- first touch on all host threads (the kernel
    // zeroes the pages), then the pin -- and the download follows it piece by piece: page zeroing, pinning and DMA
    // overlap instead of adding up (zeroing + pinning 5.3 GB take about as long as moving it over the link).
    constexpr size_t kPiece = (size_t)128 << 20;
    int *early = nullptr;
    size_t early_bytes = 0;
    std::vector<char> piece_pinned;
    std::atomic<long long> pieces_ready{0};
    std::atomic<bool> early_failed{false}, early_stop{false}, early_set{false};
    std::thread prep;
    if (bound >= 0 && bound <= INT_MAX) {
        early_bytes = (size_t)(bound > 0 ? bound : 1) * sizeof(int);
        piece_pinned.assign((early_bytes + kPiece - 1) / kPiece, 0);
        const int dev = ctx->device;
        prep = std::thread([&, dev] {
            early = static_cast<int *>(malloc(early_bytes));
            if (!early) { early_failed = true; return; }
            early_set.store(true, std::memory_order_release);
            const bool pin = early_bytes >= ((size_t)1 << 20) && hipSetDevice(dev) == hipSuccess;
            {   // huge pages for the page-aligned interior (no effect where THP is off)
                const uintptr_t lo = (reinterpret_cast<uintptr_t>(early) + ((size_t)2 << 20) - 1) & ~(((uintptr_t)2 << 20) - 1);
                const uintptr_t hi = (reinterpret_cast<uintptr_t>(early) + early_bytes) & ~(((uintptr_t)2 << 20) - 1);
                if (hi > lo) madvise(reinterpret_cast<void *>(lo), hi - lo, MADV_HUGEPAGE);
            }
            char *base = reinterpret_cast<char *>(early);
            for (size_t k = 0; k < piece_pinned.size() && !early_stop; k++) {
                const size_t off = k * kPiece, len = (early_bytes - off < kPiece) ? early_bytes - off : kPiece;
                bspgemm_par_prefault(base + off, len);
                if (pin && hipHostRegister(base + off, len, hipHostRegisterDefault) == hipSuccess) piece_pinned[k] = 1;
                else (void)hipGetLastError();
                pieces_ready.store((long long)k + 1, std::memory_order_release);
            }
        });
    }
    const double t1 = now();
    bspgemm_matrix *A = nullptr, *B = nullptr, *Fm = nullptr;
    bspgemm_result *C = nullptr;
    bspgemm_status st = bspgemm_matrix_upload(ctx, rows, brows, Arow + r0, Acol, &A);
    // A * A through the same host arrays (the reference's own call, :322): B is a view of A's device copy
    if (!st && Bcol == Acol && Brow == Arow && r0 == 0 && r1 >= brows)
        st = bspgemm_matrix_wrap_device(ctx, brows, Bm, (long long)Brow[brows] - Brow[0], A->d_row_ptr, A->d_col_idx, &B);
    else if (!st)
        st = bspgemm_matrix_upload(ctx, brows, Bm, Brow, Bcol, &B);
    if (!st && Frow) st = bspgemm_matrix_upload(ctx, rows, Bm, Frow + r0, Fcol, &Fm);
    const double t2 = now();
    if (!st) st = Fm ? bspgemm_multiply_masked(ctx, A, B, Fm, 0, rows, &C) : bspgemm_multiply(ctx, A, B, 0, rows, &C);
    const double t3 = now();
    auto unpin_early = [&] {
        char *base = reinterpret_cast<char *>(early);
        for (size_t k = 0; k < piece_pinned.size(); k++)
            if (piece_pinned[k]) { (void)hipHostUnregister(base + k * kPiece); piece_pinned[k] = 0; }
    };
    auto drop_early = [&] {                                // (the helper has been joined)
        if (!early) return;
        unpin_early();
        free(early);
        early = nullptr;
    };
    // col_idx into the early destination, piece by piece behind the helper; returns false if that cannot be used
    auto download_early = [&](long long nnz, int64_t *rp64) -> bspgemm_status {
        const size_t need = (size_t)nnz * sizeof(int);
        hipStream_t s = ctx->stream;
        HIPCHK(hipMemcpyAsync(rp64, C->d_row_ptr, ((size_t)rows + 1) * sizeof(long long), hipMemcpyDeviceToHost, s));
        for (size_t k = 0, off = 0; off < need; k++, off += kPiece) {
            while (pieces_ready.load(std::memory_order_acquire) <= (long long)k) usleep(50);
            const size_t len = (need - off < kPiece) ? need - off : kPiece;
            HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(early) + off, reinterpret_cast<const char *>(C->d_col_idx) + off, len,
                                  hipMemcpyDeviceToHost, s));
        }
        early_stop = true;                                 // pieces beyond nnz(C) are not needed
        HIPCHK(hipStreamSynchronize(s));
        return BSPGEMM_OK;
    };
    if (prep.joinable() && st) { early_stop = true; prep.join(); }
    if (!st) {
        const long long nnz = bspgemm_result_nnz(C);
        if (nnz > INT_MAX) {
            st = FAIL(BSPGEMM_ERR_OVERFLOW, "nnz(C) > INT_MAX: use the int64 handle API");
        } else {
            int *dst = nullptr;
            // the helper has malloc'ed (or failed) long before the multiply is over: wait for that one pointer
            if (prep.joinable()) while (!early_set.load(std::memory_order_acquire) && !early_failed) usleep(50);
            const bool use_early = mode == 0 && early && nnz <= bound;
            if (mode == 0) {
                if (use_early) {
                    dst = early;
                } else {
                    if (prep.joinable()) { early_stop = true; prep.join(); }
                    drop_early();
                    dst = static_cast<int *>(malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int)));
                    advise_huge(dst, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int));
                }
            } else if (mode == 1) {
                dst = *Ccol;
                if (!dst || !Csize || *Csize < nnz) {
                    dst = static_cast<int *>(realloc(*Ccol, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int)));
                    if (dst) {                          // published at once: realloc has freed or moved the old block
                        *Ccol = dst;
                        if (Csize) *Csize = (int)(nnz > 0 ? nnz : 1);
                    }
                    advise_huge(dst, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int));
                }
            } else {
                dst = *Ccol;
            }
            int64_t *rp64 = static_cast<int64_t *>(malloc(((size_t)rows + 1) * sizeof(int64_t)));
            if (!dst || !rp64) {
                st = FAIL(BSPGEMM_ERR_ALLOC, "host result");
                if (mode == 0 && !use_early) free(dst);
            } else {
                st = use_early ? download_early(nnz, rp64) : bspgemm_result_download(ctx, C, rp64, dst);
                if (use_early) {
                    early_stop = true;
                    prep.join();
                    unpin_early();
                    early = nullptr;                    // handed to the caller (or freed just below)
                    // give back what the bound overshot (in place: the block only shrinks)
                    if (!st && bound - nnz > (1 << 20)) {
                        int *shrunk = static_cast<int *>(realloc(dst, (size_t)(nnz > 0 ? nnz : 1) * sizeof(int)));
                        if (shrunk) dst = shrunk;
                    }
                }
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
    if (prep.joinable()) { early_stop = true; prep.join(); }
    drop_early();
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
// One protocol, two transports.  Every rank contributes the int32 LENGTHS of its shard's rows,
// padded to the longest shard; one all-gather; every rank scans the gathered lengths into the
// global int64 C.row_ptr on its device (bspgemm_lengths_to_row_ptr).  The all-gather is RCCL over
// xGMI (bspgemm_comm_create) or a host callback (bspgemm_comm_create_host: MPI_Allgather in the C
// drivers when ranks share a GPU, a fake in the tests) -- everything around it is the same code.
struct bspgemm_comm {
    bspgemm_context *ctx;
    ncclComm_t comm;                    // RCCL transport (NULL with the host transport)
    bspgemm_host_transport host;        // host transport callbacks (allgather NULL with RCCL)
    int rank, nranks;
    int *d_send = nullptr;              // width ints
    int *d_recv = nullptr;              // nranks * width ints
    size_t width_cap = 0;
    long long *d_global = nullptr;      // stitched row_ptr, grown on demand
    size_t global_cap = 0;
    long long *d_edges = nullptr;       // global row_ptr at the shard bounds (nranks + 1)
    int *d_bounds = nullptr;            // nranks + 1
    int *d_status = nullptr;            // 1 + nranks ints: this rank's status, everybody's (bspgemm_comm_agree)
    double timeout_s = 120.0;           // BSPGEMM_COMM_TIMEOUT_S: a collective that has not completed by then is aborted
    int inject = 0;                     // bspgemm_comm_inject_failure (tests)
};

// Waits for the stream behind an RCCL call without trusting it to finish: a peer that died or left the
// protocol leaves the others inside the collective for ever (the reference's MPI calls have the same
// property: final/SpGEMM_mpi_omp.c:178-204 checks nothing).  Polls the stream and RCCL's asynchronous
// error state; on an error or after timeout_s the communicator is aborted and the call FAILS.
static bspgemm_status comm_wait(bspgemm_comm *c, hipStream_t s, const char *what)
{
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
        const hipError_t q = hipStreamQuery(s);
        if (q == hipSuccess) return BSPGEMM_OK;
        if (q != hipErrorNotReady) {
            snprintf(g_err, sizeof g_err, "%s: %s", what, hipGetErrorString(q));
            return BSPGEMM_ERR_HIP;
        }
        ncclResult_t async = ncclSuccess;
        if (c->comm && ncclCommGetAsyncError(c->comm, &async) == ncclSuccess && async != ncclSuccess && async != ncclInProgress) {
            snprintf(g_err, sizeof g_err, "%s: RCCL reported %s; communicator aborted", what, ncclGetErrorString(async));
            ncclCommAbort(c->comm);
            c->comm = nullptr;
            return BSPGEMM_ERR_COMM;
        }
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > c->timeout_s) {
            snprintf(g_err, sizeof g_err, "%s: no completion after %.0f s (a peer is missing?); communicator aborted", what, c->timeout_s);
            if (c->comm) { ncclCommAbort(c->comm); c->comm = nullptr; }
            return BSPGEMM_ERR_COMM;
        }
        usleep(200);
    }
}

extern "C" bspgemm_status bspgemm_comm_unique_id(unsigned char id[BSPGEMM_UNIQUE_ID_BYTES])
{
    static_assert(sizeof(ncclUniqueId) == BSPGEMM_UNIQUE_ID_BYTES, "RCCL unique id size");
    if (!id) return FAIL(BSPGEMM_ERR_INVALID, "id is NULL");
    ncclUniqueId u;
    NCCLCHK(ncclGetUniqueId(&u));
    memcpy(id, &u, sizeof u);
    return BSPGEMM_OK;
}

static bspgemm_status comm_new(bspgemm_context *ctx, int rank, int nranks, bspgemm_comm **out)
{
    bspgemm_comm *c = new (std::nothrow) bspgemm_comm();
    if (!c) return FAIL(BSPGEMM_ERR_ALLOC, "comm");
    c->ctx = ctx;
    c->comm = nullptr;
    c->host = bspgemm_host_transport{nullptr, nullptr, nullptr};
    c->rank = rank;
    c->nranks = nranks;
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&c->d_edges), ((size_t)nranks + 1) * sizeof(long long));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->d_bounds), ((size_t)nranks + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&c->d_status), ((size_t)nranks + 1) * sizeof(int));
    if (const char *t = getenv("BSPGEMM_COMM_TIMEOUT_S")) { const double v = atof(t); if (v > 0) c->timeout_s = v; }
    if (e != hipSuccess) {
        bspgemm_comm_destroy(c);
        snprintf(g_err, sizeof g_err, "comm buffers: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? BSPGEMM_ERR_ALLOC : BSPGEMM_ERR_HIP;
    }
    *out = c;
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_comm_create(bspgemm_context *ctx, const unsigned char id[BSPGEMM_UNIQUE_ID_BYTES],
                                              int rank, int nranks, bspgemm_comm **out)
{
    if (!ctx || !id || !out || nranks <= 0 || rank < 0 || rank >= nranks) return FAIL(BSPGEMM_ERR_INVALID, "comm_create");
    *out = nullptr;
    if (bspgemm_status st = use_device(ctx)) return st;
    bspgemm_comm *c = nullptr;
    if (bspgemm_status st = comm_new(ctx, rank, nranks, &c)) return st;
    ncclUniqueId u;
    memcpy(&u, id, sizeof u);
    // RCCL prints a version banner on STDOUT when the first communicator is created; the drop-in
    // binaries' stdout is the reference's CSV line / parity message and nothing else, so the
    // banner is sent to stderr
    fflush(stdout);
    const int saved_out = dup(1);
    if (saved_out >= 0) dup2(2, 1);
    ncclResult_t r = ncclCommInitRank(&c->comm, nranks, u, rank);
    if (saved_out >= 0) {
        fflush(stdout);
        dup2(saved_out, 1);
        close(saved_out);
    }
    if (r != ncclSuccess) {
        c->comm = nullptr;
        bspgemm_comm_destroy(c);
        snprintf(g_err, sizeof g_err, "ncclCommInitRank: %s", ncclGetErrorString(r));
        return BSPGEMM_ERR_COMM;
    }
    *out = c;
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_comm_create_host(bspgemm_context *ctx, const bspgemm_host_transport *t,
                                                   int rank, int nranks, bspgemm_comm **out)
{
    if (!ctx || !t || !t->allgather || !out || nranks <= 0 || rank < 0 || rank >= nranks)
        return FAIL(BSPGEMM_ERR_INVALID, "comm_create_host");
    *out = nullptr;
    if (bspgemm_status st = use_device(ctx)) return st;
    bspgemm_comm *c = nullptr;
    if (bspgemm_status st = comm_new(ctx, rank, nranks, &c)) return st;
    c->host = *t;
    *out = c;
    return BSPGEMM_OK;
}

extern "C" void bspgemm_comm_destroy(bspgemm_comm *c)
{
    if (!c) return;
    hipSetDevice(c->ctx->device);
    if (c->comm) ncclCommDestroy(c->comm);
    hipFree(c->d_send);
    hipFree(c->d_recv);
    hipFree(c->d_global);
    hipFree(c->d_edges);
    hipFree(c->d_bounds);
    hipFree(c->d_status);
    delete c;
}

extern "C" void bspgemm_comm_inject_failure(bspgemm_comm *c, int what) { if (c) c->inject = what; }

// Every rank contributes its status; everybody gets the worst.  Called before a collective that a failed
// rank would not enter: either all ranks go in, or none does (a rank that left the protocol alone leaves
// the others blocked in ncclSend / MPI_Gatherv for ever).  Collective.
extern "C" bspgemm_status bspgemm_comm_agree(bspgemm_comm *c, bspgemm_status mine)
{
    if (!c) return FAIL(BSPGEMM_ERR_INVALID, "comm is NULL");
    bspgemm_context *ctx = c->ctx;
    if (bspgemm_status st = use_device(ctx)) return st;
    hipStream_t s = ctx->stream;
    const int n = c->nranks;
    int worst = (int)mine;
    if (c->comm) {
        const int v = (int)mine;
        HIPCHK(hipMemcpyAsync(c->d_status, &v, sizeof(int), hipMemcpyHostToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        NCCLCHK(ncclAllGather(c->d_status, c->d_status + 1, 1, ncclInt32, c->comm, s));
        if (bspgemm_status st = comm_wait(c, s, "status all-gather")) return st;
        int all[1025];
        if (n > 1024) return FAIL(BSPGEMM_ERR_INVALID, "more than 1024 ranks");
        HIPCHK(hipMemcpy(all, c->d_status + 1, (size_t)n * sizeof(int), hipMemcpyDeviceToHost));
        for (int r = 0; r < n; r++) if (all[r] > worst) worst = all[r];
    } else {
        const int v = (int)mine;
        int *all = static_cast<int *>(malloc((size_t)n * sizeof(int)));
        if (!all) return FAIL(BSPGEMM_ERR_ALLOC, "status staging");
        const int rc = c->host.allgather(c->host.user, &v, all, sizeof(int));
        if (rc == 0) for (int r = 0; r < n; r++) if (all[r] > worst) worst = all[r];
        free(all);
        if (rc != 0) return FAIL(BSPGEMM_ERR_COMM, "host all-gather failed");
    }
    if (worst != BSPGEMM_OK && mine == BSPGEMM_OK)
        snprintf(g_err, sizeof g_err, "another rank failed with status %d (%s)", worst, bspgemm_status_string((bspgemm_status)worst));
    return (bspgemm_status)worst;
}

extern "C" int bspgemm_comm_rank(const bspgemm_comm *c) { return c ? c->rank : -1; }
extern "C" int bspgemm_comm_size(const bspgemm_comm *c) { return c ? c->nranks : 0; }

// lengths[i] = row_ptr[i+1] - row_ptr[i]  (|C_i| < 2^31 always; the slice-local offsets are not)
__global__ void k_row_lengths(const long long *__restrict__ row_ptr, int n, int *__restrict__ len)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) len[i] = (int)(row_ptr[i + 1] - row_ptr[i]);
}
// edges[r] = global[bounds[r]]
__global__ void k_pick_edges(const long long *__restrict__ global, const int *__restrict__ bounds, int n,
                             long long *__restrict__ edges)
{
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) edges[r] = global[bounds[r]];
}

static bspgemm_status comm_bounds_ok(const bspgemm_comm *c, const int *bounds)
{
    if (bounds[0] != 0) return FAIL(BSPGEMM_ERR_INVALID, "bounds[0] != 0");
    for (int r = 0; r < c->nranks; r++)
        if (bounds[r + 1] < bounds[r]) return FAIL(BSPGEMM_ERR_INVALID, "bounds not ascending");
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_comm_stitch_row_ptr(bspgemm_comm *c, const bspgemm_result *local,
                                                      const int *bounds, const int64_t **d_row_ptr_global,
                                                      int64_t *shard_nnz)
{
    if (!c || !local || !bounds || !d_row_ptr_global) return FAIL(BSPGEMM_ERR_INVALID, "stitch_row_ptr");
    if (bspgemm_status st = comm_bounds_ok(c, bounds)) return st;
    if (c->nranks > 1024) return FAIL(BSPGEMM_ERR_INVALID, "more than 1024 ranks");
    const int my_rows = bounds[c->rank + 1] - bounds[c->rank];
    if (my_rows != local->rows) return FAIL(BSPGEMM_ERR_INVALID, "local result does not match bounds[rank]");
    bspgemm_context *ctx = c->ctx;
    if (bspgemm_status st = use_device(ctx)) return st;
    hipStream_t s = ctx->stream;
    int width = 1;
    for (int r = 0; r < c->nranks; r++)
        if (bounds[r + 1] - bounds[r] > width) width = bounds[r + 1] - bounds[r];
    const size_t need = (size_t)bounds[c->nranks] + 1;
    if ((size_t)width > c->width_cap || need > c->global_cap) {
        HIPCHK(hipStreamSynchronize(s));
        if ((size_t)width > c->width_cap) {
            hipFree(c->d_send); hipFree(c->d_recv);
            c->d_send = c->d_recv = nullptr;
            c->width_cap = 0;
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_send), (size_t)width * sizeof(int)));
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_recv), (size_t)width * c->nranks * sizeof(int)));
            c->width_cap = (size_t)width;
        }
        if (need > c->global_cap) {
            hipFree(c->d_global);
            c->d_global = nullptr;
            c->global_cap = 0;
            HIPCHK(hipMalloc(reinterpret_cast<void **>(&c->d_global), need * sizeof(long long)));
            c->global_cap = need;
        }
    }
    // 1. this shard's row lengths (pad slots are never read by the scan)
    if (my_rows > 0)
        hipLaunchKernelGGL(k_row_lengths, dim3((my_rows + 255) / 256), dim3(256), 0, s, local->d_row_ptr, my_rows, c->d_send);
    // 2. the one collective
    if (c->comm) {
        NCCLCHK(ncclAllGather(c->d_send, c->d_recv, (size_t)width, ncclInt32, c->comm, s));
        if (bspgemm_status st = comm_wait(c, s, "row-length all-gather")) return st;
    } else {
        const size_t bytes = (size_t)width * sizeof(int);
        int *hs = static_cast<int *>(malloc(bytes)), *hr = static_cast<int *>(malloc(bytes * c->nranks));
        bspgemm_status st = (hs && hr) ? BSPGEMM_OK : FAIL(BSPGEMM_ERR_ALLOC, "host staging");
        if (!st && hipMemcpyAsync(hs, c->d_send, bytes, hipMemcpyDeviceToHost, s) != hipSuccess) st = FAIL(BSPGEMM_ERR_HIP, "lengths to host");
        if (!st && hipStreamSynchronize(s) != hipSuccess) st = FAIL(BSPGEMM_ERR_HIP, "sync");
        if (!st && c->host.allgather(c->host.user, hs, hr, bytes) != 0) st = FAIL(BSPGEMM_ERR_COMM, "host all-gather failed");
        if (!st && hipMemcpyAsync(c->d_recv, hr, bytes * c->nranks, hipMemcpyHostToDevice, s) != hipSuccess) st = FAIL(BSPGEMM_ERR_HIP, "lengths to device");
        if (!st && hipStreamSynchronize(s) != hipSuccess) st = FAIL(BSPGEMM_ERR_HIP, "sync");
        free(hs); free(hr);
        if (st) return st;
    }
    // 3. every rank scans the gathered lengths: shard r continues where shard r-1 ended
    if (bspgemm_status st = bspgemm_lengths_to_row_ptr(ctx, c->d_recv, c->nranks, width, bounds,
                                                       reinterpret_cast<int64_t *>(c->d_global), s))
        return st;
    *d_row_ptr_global = reinterpret_cast<const int64_t *>(c->d_global);
    if (shard_nnz) {
        HIPCHK(hipMemcpyAsync(c->d_bounds, bounds, ((size_t)c->nranks + 1) * sizeof(int), hipMemcpyHostToDevice, s));
        hipLaunchKernelGGL(k_pick_edges, dim3(1), dim3(c->nranks + 1 <= 1024 ? c->nranks + 1 : 1024), 0, s,
                           c->d_global, c->d_bounds, c->nranks + 1, c->d_edges);
        long long edges[1025];
        HIPCHK(hipMemcpyAsync(edges, c->d_edges, ((size_t)c->nranks + 1) * sizeof(long long), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        for (int r = 0; r < c->nranks; r++) shard_nnz[r] = edges[r + 1] - edges[r];
    } else {
        HIPCHK(hipStreamSynchronize(s));
    }
    return BSPGEMM_OK;
}

// Root gather of the sharded col_idx (what MPI_Gatherv does at final/SpGEMM_mpi_omp.c:203): with
// RCCL every rank sends its shard and the root receives them at their global offsets -- one
// grouped send/recv, the root's own shard included, so that one rank runs the same calls as N.
extern "C" bspgemm_status bspgemm_comm_gather_col_idx(bspgemm_comm *c, const bspgemm_result *local,
                                                      const int64_t *shard_nnz, int root, int *col_idx_host)
{
    if (!c || !local || !shard_nnz || root < 0 || root >= c->nranks) return FAIL(BSPGEMM_ERR_INVALID, "gather_col_idx");
    if (shard_nnz[c->rank] != local->nnz) return FAIL(BSPGEMM_ERR_INVALID, "shard_nnz[rank] != nnz of the local result");
    bspgemm_context *ctx = c->ctx;
    if (bspgemm_status st = use_device(ctx)) return st;
    hipStream_t s = ctx->stream;
    long long total = 0;
    for (int r = 0; r < c->nranks; r++) total += shard_nnz[r];
    // A root WITHOUT a destination (its malloc failed) still runs the whole collective -- the peers are already
    // committed to it -- and reports the failure afterwards: an early return here left them blocked in ncclSend /
    // MPI_Gatherv.  (SpGEMM_hip_multi agrees on the status first, so this is the second line of defence.)
    const bool root_blind = c->rank == root && total > 0 && !col_idx_host;
    if (c->comm) {
        int *d_all = nullptr;
        if (c->rank == root)
            HIPCHK(result_alloc(ctx, reinterpret_cast<void **>(&d_all), result_bytes_colidx(total)));
        ncclResult_t nr = ncclGroupStart();
        if (nr == ncclSuccess && local->nnz > 0)
            nr = ncclSend(local->d_col_idx, (size_t)local->nnz, ncclInt32, root, c->comm, s);
        if (c->rank == root) {
            long long off = 0;
            for (int r = 0; r < c->nranks && nr == ncclSuccess; r++) {
                if (shard_nnz[r] > 0) nr = ncclRecv(d_all + off, (size_t)shard_nnz[r], ncclInt32, r, c->comm, s);
                off += shard_nnz[r];
            }
        }
        const ncclResult_t ne = ncclGroupEnd();
        if (nr == ncclSuccess) nr = ne;
        bspgemm_status st = BSPGEMM_OK;
        if (nr != ncclSuccess) {
            snprintf(g_err, sizeof g_err, "col_idx gather: %s", ncclGetErrorString(nr));
            st = BSPGEMM_ERR_COMM;
        }
        if (!st) st = comm_wait(c, s, "col_idx gather");
        if (!st && c->rank == root && total > 0 && !root_blind &&
            hipMemcpyAsync(col_idx_host, d_all, (size_t)total * sizeof(int), hipMemcpyDeviceToHost, s) != hipSuccess)
            st = FAIL(BSPGEMM_ERR_HIP, "gathered col_idx to host");
        if (hipStreamSynchronize(s) != hipSuccess && !st) st = FAIL(BSPGEMM_ERR_HIP, "sync");
        if (d_all) result_release(ctx, d_all, result_bytes_colidx(total));
        if (!st && root_blind) st = FAIL(BSPGEMM_ERR_INVALID, "root has no destination (the gather was run and discarded)");
        return st;
    }
    if (!c->host.gatherv) return FAIL(BSPGEMM_ERR_INVALID, "host transport has no gatherv");
    int *mine = static_cast<int *>(malloc((size_t)(local->nnz > 0 ? local->nnz : 1) * sizeof(int)));
    if (!mine) return FAIL(BSPGEMM_ERR_ALLOC, "host shard");
    bspgemm_status st = bspgemm_result_download(ctx, local, nullptr, mine);
    if (!st) {
        size_t *bytes = static_cast<size_t *>(malloc((size_t)c->nranks * sizeof(size_t)));
        if (!bytes) st = FAIL(BSPGEMM_ERR_ALLOC, "counts");
        else {
            for (int r = 0; r < c->nranks; r++) bytes[r] = (size_t)shard_nnz[r] * sizeof(int);
            int *scratch = nullptr;                          // a blind root receives into scratch and discards
            if (root_blind) scratch = static_cast<int *>(malloc((size_t)total * sizeof(int)));
            if (c->host.gatherv(c->host.user, mine, (size_t)local->nnz * sizeof(int), root_blind ? scratch : col_idx_host, bytes, root) != 0)
                st = FAIL(BSPGEMM_ERR_COMM, "host gatherv failed");
            free(scratch);
            if (!st && root_blind) st = FAIL(BSPGEMM_ERR_INVALID, "root has no destination (the gather was run and discarded)");
            free(bytes);
        }
    }
    free(mine);
    return st;
}

// replaces SpGEMM_mpi (final/SpGEMM_mpi_omp.c:155-225): same arguments plus the communicator the
// reference takes implicitly (MPI_COMM_WORLD).  Every rank passes the whole A and B (every rank of
// the reference reads the whole file, :309); rows are cut at equal work instead of An/numtasks
// (:165); the result -- malloc'ed *Ccol, caller's Crow[An+1] -- is valid on rank 0 only, like :200-223.
extern "C" int SpGEMM_hip_multi(bspgemm_comm *c, int *Acol, int *Arow, int An, int *Bcol, int *Brow, int Bm,
                                int **Ccol, int *Crow, int tBlock)
{
    (void)tBlock;
    if (Ccol) *Ccol = nullptr;
    auto run = [&]() -> bspgemm_status {
        if (!c || !Acol || !Arow || !Bcol || !Brow || !Ccol || !Crow || An < 0 || Bm < 0)
            return FAIL(BSPGEMM_ERR_INVALID, "SpGEMM_hip_multi arguments");
        bspgemm_context *ctx = c->ctx;
        const int brows = bspgemm_par_max_plus_one(Acol + Arow[0], (long long)Arow[An] - Arow[0]);
        bspgemm_matrix *A = nullptr, *B = nullptr;
        bspgemm_result *C = nullptr;
        int *bounds = static_cast<int *>(malloc(((size_t)c->nranks + 1) * sizeof(int)));
        int64_t *shard = static_cast<int64_t *>(malloc((size_t)c->nranks * sizeof(int64_t)));
        int64_t *rp64 = nullptr;
        int *dst = nullptr;
        bspgemm_status st = (bounds && shard) ? BSPGEMM_OK : FAIL(BSPGEMM_ERR_ALLOC, "bounds");
        if (!st) st = bspgemm_matrix_upload(ctx, An, brows, Arow, Acol, &A);
        if (!st) st = bspgemm_matrix_upload(ctx, brows, Bm, Brow, Bcol, &B);
        if (!st) st = bspgemm_partition_rows(ctx, A, B, c->nranks, bounds);
        if (!st) st = bspgemm_multiply(ctx, A, B, bounds[c->rank], bounds[c->rank + 1], &C);
        // from here on the ranks act together: nobody enters a collective unless everybody does
        st = bspgemm_comm_agree(c, st);
        const int64_t *d_global = nullptr;
        if (!st) st = bspgemm_comm_stitch_row_ptr(c, C, bounds, &d_global, shard);
        st = bspgemm_comm_agree(c, st);
        long long total = 0;
        if (!st) {
            for (int r = 0; r < c->nranks; r++) total += shard[r];
            if (total > INT_MAX) st = FAIL(BSPGEMM_ERR_OVERFLOW, "nnz(C) > INT_MAX: use the int64 handle API");
        }
        if (!st && c->rank == 0) {
            if (c->inject != 1) dst = static_cast<int *>(malloc((size_t)(total > 0 ? total : 1) * sizeof(int)));
            rp64 = static_cast<int64_t *>(malloc(((size_t)An + 1) * sizeof(int64_t)));
            if (!dst || !rp64) st = FAIL(BSPGEMM_ERR_ALLOC, "host result");
        }
        // the root's allocation is the last thing that can fail on one rank only: agreed on before the gather, so
        // that either every rank enters it or none does
        st = bspgemm_comm_agree(c, st);
        if (!st) st = bspgemm_comm_gather_col_idx(c, C, shard, 0, dst);
        if (!st && c->rank == 0) {
            if (hipMemcpy(rp64, d_global, ((size_t)An + 1) * sizeof(int64_t), hipMemcpyDeviceToHost) != hipSuccess)
                st = FAIL(BSPGEMM_ERR_HIP, "row_ptr to host");
            else {
                for (int i = 0; i <= An; i++) Crow[i] = (int)rp64[i];
                *Ccol = dst;
                dst = nullptr;
            }
        }
        free(dst); free(rp64); free(bounds); free(shard);
        bspgemm_result_free(C);
        bspgemm_matrix_free(A);
        bspgemm_matrix_free(B);
        return st;
    };
    const bspgemm_status st = run();
    return st ? dropin_fail("SpGEMM_hip_multi", st) : 0;
}
