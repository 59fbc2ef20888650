// context.hip -- the handle side of the C ABI (include/bspgemm.h): errors, the per-GPU context and its
// workspaces, device-resident operands and their derived tables, the cache of freed result buffers,
// result accessors, statistics.  No CPU compute path exists in this library: without a gfx950 device
// every compute entry point fails with BSPGEMM_ERR_NO_DEVICE.
#include "internal.hpp"

using namespace bsp;

// ------------------------------------------------------------------ errors ---------------
thread_local char bspgemm_err_text[512] = "";

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


bspgemm_status ensure_deg8(const bspgemm_matrix *m)
{
    if (m->d_deg8) return BSPGEMM_OK;
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->d_deg8), (size_t)m->rows + 1));
    launch_deg8(m->d_row_ptr, m->rows, m->d_deg8, m->ctx->stream);
    HIPCHK(hipGetLastError());
    return BSPGEMM_OK;
}

// The padded copy of B.col_idx (rows on 64-byte boundaries).  OPT-IN (BSPGEMM_OPT_PADDED_ROWS, default off): it cuts the
// bytes the gather moves by 22 % on the bench matrix (9.4 -> 7.3 GB, tools/gather_traffic_model.py) and the numeric phase by
// nothing there (3.336 -> 3.327 / 3.314 ms, profiles/r04_ab_padded_rows.log); with rows of exactly 16 entries (uniform
// matrices: every row is one aligned sector) it is worth 8-9 % of the numeric phase (profiles/r04_alignment_probe.log).
// Value -1 decides per operand: at least 2^20 nonzeros, a mean row length of 8 or more, the padded copy at most twice the
// original and below 2^31 entries.
bspgemm_status ensure_pad(const bspgemm_matrix *m)
{
    if (m->pad_state) return BSPGEMM_OK;
    m->pad_state = 2;
    bspgemm_context *ctx = m->ctx;
    const int force = ctx->pad_rows;
    if (force == 0 || m->rows <= 0 || m->nnz <= 0) return BSPGEMM_OK;
    if (force < 0 && (m->nnz < (1ll << 20) || m->nnz < 8ll * m->rows)) return BSPGEMM_OK;
    hipStream_t s = ctx->stream;
    const size_t rows = (size_t)m->rows;
    int *plen = nullptr;
    long long *pp64 = nullptr, *partials = nullptr;
    auto drop = [&] { hipFree(plen); hipFree(pp64); hipFree(partials); };
    auto bail = [&](bspgemm_status st) {
        hipStreamSynchronize(s);
        drop();
        hipFree(m->d_col_pad); hipFree(m->d_row_ptr_pad); hipFree(m->d_ext);
        m->d_col_pad = nullptr; m->d_row_ptr_pad = nullptr; m->d_ext = nullptr;
        return st;
    };
    HIPCHK_B(hipMalloc(reinterpret_cast<void **>(&plen), rows * sizeof(int)));
    HIPCHK_B(hipMalloc(reinterpret_cast<void **>(&pp64), (rows + 1) * sizeof(long long)));
    HIPCHK_B(hipMalloc(reinterpret_cast<void **>(&partials), (rows / 2048 + 4) * sizeof(long long)));
    launch_pad_lengths(m->d_row_ptr, m->rows, plen, s);
    launch_scan_counts(plen, m->rows, pp64, partials, nullptr, s);
    long long total = 0;
    HIPCHK_B(hipMemcpyAsync(&total, pp64 + rows, sizeof(long long), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipStreamSynchronize(s));
    if (total > 0x7fffffffll - 64 || (force < 0 && total > 2 * m->nnz)) { drop(); return BSPGEMM_OK; }   // (stays "not for this operand")
    hipError_t e = hipMalloc(reinterpret_cast<void **>(&m->d_col_pad), ((size_t)total + 64) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->d_row_ptr_pad), (rows + 1) * sizeof(int));
    if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void **>(&m->d_ext), rows * sizeof(int2));
    if (e != hipSuccess) {
        (void)hipGetLastError();
        if (force == 1) { snprintf(g_err, sizeof g_err, "padded copy of col_idx: %s", hipGetErrorString(e)); return bail(BSPGEMM_ERR_ALLOC); }
        (void)bail(BSPGEMM_OK);                                // optional: an operand that does not fit twice is simply not padded
        return BSPGEMM_OK;
    }
    launch_narrow_row_ptr(pp64, m->d_row_ptr_pad, m->rows + 1, s);
    launch_pad_copy(m->d_row_ptr, m->d_col_idx, m->d_row_ptr_pad, m->rows, m->d_col_pad, m->d_ext, s);
    HIPCHK_B(hipGetLastError());
    HIPCHK_B(hipStreamSynchronize(s));
    drop();
    m->pad_state = 1;
    return BSPGEMM_OK;
}

// Whether products with `m` as B go through the blocked table.  It pays when B.row_ptr is several
// times an XCD's 4 MB L2 (R-MAT scale 22: 16.8 MB, k_row_work 1.20 -> 0.92 ms) and the operand is not
// dominated by rows of 255+ nonzeros, whose lengths the table clamps (power-law n = 2^20: B.row_ptr
// fits L2 anyway and 10 % of the lookups fall through: 2.0 -> 3.0 ms; Graph500 skew: 70 % fall through).
// Decided once per operand: one 8-byte read-back when the table is built.
bspgemm_status ensure_blk8(const bspgemm_matrix *m)
{
    if (m->blk8_state) return BSPGEMM_OK;
    m->blk8_state = 2;
    const int force = m->ctx->rw_blk;                              // 0 never, 1 always, -1 decide per operand
    if (force == 0 || (force < 0 && m->rows < (1 << 21))) return BSPGEMM_OK;
    const size_t ints = (size_t)3 * (((size_t)m->rows + 7) / 8 + 1);
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&m->d_blk8), (ints + 4) * sizeof(int)));
    unsigned long long *d_clamped = reinterpret_cast<unsigned long long *>(m->d_blk8 + ((ints + 1) & ~(size_t)1));
    HIPCHK(hipMemsetAsync(d_clamped, 0, sizeof(unsigned long long), m->ctx->stream));
    launch_blk8(m->d_row_ptr, m->pad_state == 1 ? m->d_row_ptr_pad : nullptr, m->rows, m->d_blk8, d_clamped, m->ctx->stream);
    HIPCHK(hipGetLastError());
    unsigned long long clamped = 0;
    HIPCHK(hipMemcpyAsync(&clamped, d_clamped, sizeof(clamped), hipMemcpyDeviceToHost, m->ctx->stream));
    HIPCHK(hipStreamSynchronize(m->ctx->stream));
    if (force == 1 || clamped * 8ull <= (unsigned long long)m->nnz) m->blk8_state = 1;
    return BSPGEMM_OK;
}

bspgemm_status use_device(bspgemm_context *ctx)
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
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->d_small), sizeof(SmallScalars)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->d_small_tiles), sizeof(SmallTiles)));
    HIPCHK(hipMalloc(reinterpret_cast<void **>(&ctx->d_err), 64));
    HIPCHK(hipMemset(ctx->d_err, 0, 64));
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
        ctx->flow = !strcmp(e, "exact") ? BSPGEMM_FLOW_EXACT : (!strcmp(e, "upper-bound") || !strcmp(e, "ub")) ? BSPGEMM_FLOW_UPPER_BOUND : BSPGEMM_FLOW_AUTO;
    if (const char *e = getenv("BSPGEMM_CLASS_TIMING")) ctx->class_timing = atoi(e) != 0;
    if (const char *e = getenv("BSPGEMM_CLASS_STREAMS")) { const int n = atoi(e); ctx->class_streams = n < 1 ? 1 : (n > 3 ? 3 : n); }
    ctx->check = getenv("BSPGEMM_CHECK") != nullptr;
    if (const char *e = getenv("BSPGEMM_RW_BLK")) ctx->rw_blk = atoi(e) ? 1 : 0;
    if (const char *e = getenv("BSPGEMM_SMALL")) ctx->small = atoi(e) ? 1 : 0;
    if (const char *e = getenv("BSPGEMM_PAD_ROWS")) { const int v = atoi(e); ctx->pad_rows = v < 0 ? -1 : (v ? 1 : 0); }
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
    if (ctx->h) hipHostFree(ctx->h);
    hipFree(ctx->d_prep);
    hipFree(ctx->d_err);
    hipFree(ctx->d_small);
    hipFree(ctx->d_small_tiles);
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
    hipFree(m->d_col_pad);
    hipFree(m->d_row_ptr_pad);
    hipFree(m->d_ext);
    delete m;
}
extern "C" bspgemm_status bspgemm_matrix_invalidate(bspgemm_matrix *m)
{
    if (!m) return FAIL(BSPGEMM_ERR_INVALID, "matrix is NULL");
    if (bspgemm_status st = use_device(m->ctx)) return st;
    HIPCHK(hipStreamSynchronize(m->ctx->stream));          // a multiply may still be reading the tables
    hipFree(m->d_deg8);
    hipFree(m->d_blk8);
    hipFree(m->d_col_pad);
    hipFree(m->d_row_ptr_pad);
    hipFree(m->d_ext);
    m->d_deg8 = nullptr;
    m->d_blk8 = nullptr;
    m->blk8_state = 0;
    m->d_col_pad = nullptr;
    m->d_row_ptr_pad = nullptr;
    m->d_ext = nullptr;
    m->pad_state = 0;
    return BSPGEMM_OK;
}

extern "C" const char *bspgemm_build_info(void)
{
    return "libbspgemm: HIP kernels for gfx950 only; flows upper-bound (default), exact; small-product path; "
           "timing-only ablation switches: none (BSP_ABLATE=0); tuning constants are compile-time";
}

extern "C" int bspgemm_matrix_rows(const bspgemm_matrix *m) { return m ? m->rows : 0; }
extern "C" int bspgemm_matrix_cols(const bspgemm_matrix *m) { return m ? m->cols : 0; }
extern "C" int64_t bspgemm_matrix_nnz(const bspgemm_matrix *m) { return m ? m->nnz : 0; }

// ------------------------------------------------------------------ workspace ------------
bspgemm_status ensure_rows(bspgemm_context *ctx, size_t rows)
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

bspgemm_status ensure_ab(bspgemm_context *ctx, size_t pairs)
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

bspgemm_status ensure_tmp(bspgemm_context *ctx, size_t ints)
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

bspgemm_status ensure_chunk_rows(bspgemm_context *ctx, size_t entries)
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
bool result_cached(const bspgemm_context *ctx, size_t bytes) { return result_cache_find(ctx, bytes) >= 0; }

hipError_t result_alloc(bspgemm_context *ctx, void **out, size_t bytes)
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

void result_release(bspgemm_context *ctx, void *p, size_t bytes)
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
        st.bin_cap[b] = b == 0 ? 0 : (b == kDenseBin ? 0x7fffffff : (b == kMidBin ? sl.mid_cap : (b == kRankBin ? (sl.rank_cap > kMaxWaveCap ? sl.rank_cap : kMaxWaveCap) : 64 * kWaveChunks[b])));
    }
    st.flow = sl.flow;
    st.prepass_kernel = sl.prepass_kernel;
    st.class_streams = sl.class_streams;
    st.small_path = sl.small ? 1 : 0;
    st.checked = sl.checked ? 1 : 0;
    st.padded_rows = sl.padded ? 1 : 0;
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

extern "C" bspgemm_status bspgemm_set_class_timing(bspgemm_context *ctx, int on)
{
    if (!ctx) return FAIL(BSPGEMM_ERR_INVALID, "set_class_timing");
    ctx->class_timing = on != 0;
    return BSPGEMM_OK;
}

extern "C" bspgemm_status bspgemm_set_option(bspgemm_context *ctx, bspgemm_option opt, int value)
{
    if (!ctx) return FAIL(BSPGEMM_ERR_INVALID, "set_option: ctx is NULL");
    switch (opt) {
    case BSPGEMM_OPT_CLASS_STREAMS:
        if (value < 1 || value > 3) return FAIL(BSPGEMM_ERR_INVALID, "class streams: 1..3");
        ctx->class_streams = value;
        return BSPGEMM_OK;
    case BSPGEMM_OPT_BLOCKED_EXTENTS:
        if (value < -1 || value > 1) return FAIL(BSPGEMM_ERR_INVALID, "blocked extents: -1, 0 or 1");
        ctx->rw_blk = value;
        return BSPGEMM_OK;
    case BSPGEMM_OPT_CHECK:
        ctx->check = value != 0;
        return BSPGEMM_OK;
    case BSPGEMM_OPT_SMALL_PATH:
        if (value < -1 || value > 1) return FAIL(BSPGEMM_ERR_INVALID, "small path: -1, 0 or 1");
        ctx->small = value;
        return BSPGEMM_OK;
    case BSPGEMM_OPT_PADDED_ROWS:
        if (value < -1 || value > 1) return FAIL(BSPGEMM_ERR_INVALID, "padded rows: -1, 0 or 1");
        ctx->pad_rows = value;
        return BSPGEMM_OK;
    }
    return FAIL(BSPGEMM_ERR_INVALID, "unknown option");
}

extern "C" int bspgemm_get_option(const bspgemm_context *ctx, bspgemm_option opt)
{
    if (!ctx) return INT_MIN;
    switch (opt) {
    case BSPGEMM_OPT_CLASS_STREAMS: return ctx->class_streams;
    case BSPGEMM_OPT_BLOCKED_EXTENTS: return ctx->rw_blk;
    case BSPGEMM_OPT_CHECK: return ctx->check ? 1 : 0;
    case BSPGEMM_OPT_SMALL_PATH: return ctx->small;
    case BSPGEMM_OPT_PADDED_ROWS: return ctx->pad_rows;
    }
    return INT_MIN;
}

extern "C" int bspgemm_matrix_uses_padded_rows(const bspgemm_matrix *m)
{
    if (!m || m->pad_state == 0) return -1;
    return m->pad_state == 1 ? 1 : 0;
}

extern "C" int bspgemm_matrix_uses_blocked_table(const bspgemm_matrix *m)
{
    if (!m || m->blk8_state == 0) return -1;
    return m->blk8_state == 1 ? 1 : 0;
}

extern "C" bspgemm_status bspgemm_set_flow(bspgemm_context *ctx, int flow)
{
    if (!ctx || flow < BSPGEMM_FLOW_AUTO || flow > BSPGEMM_FLOW_EXACT) return FAIL(BSPGEMM_ERR_INVALID, "set_flow");
    ctx->flow = flow;
    return BSPGEMM_OK;
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

