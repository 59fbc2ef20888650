// multiply.hip -- bspgemm_multiply and what is built on it: the two ways from row sizes to C.col_idx
// (upper-bound placement + compaction, exact symbolic sizes + emit in place), the masked product, products
// as operands, the closure, the sharding helpers.  Replaces SpGEMM_omp / SpGEMM_bigslice
// (final/SpGEMM_mpi_omp.c:71-143, :15-58) behind the native handle API.
#include "internal.hpp"

using namespace bsp;

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
static void class_order(const int *bin_count, long long total_products, int *order, int *lane)
{
    int light[kWaveBins];                                         // the one-wave classes in launch order
    for (int k = 0; k < kWaveBins; k++) light[k] = k + 1;
    const long long heavy_lower_bound = ((long long)bin_count[kRankBin] + bin_count[kMidBin] + bin_count[kDenseBin]) * kMaxWaveCap;
    if (heavy_lower_bound * 8 < total_products) {
        long long key[kNumBins] = {};
        for (int b = 1; b <= kWaveBins; b++) key[b] = (long long)bin_count[b] * kWaveChunks[b];
        for (int a = 1; a < kWaveBins; a++)                       // insertion sort of 16 entries, stable
            for (int c = a; c > 0 && key[light[c]] > key[light[c - 1]]; c--) {
                const int t = light[c];
                light[c] = light[c - 1];
                light[c - 1] = t;
            }
    }
    // streams (taken modulo the number in use): the hub rows on 1, the small dense shape on 0 and the rank class behind IT --
    // the hub rows' stream ends last where heavy rows matter (power-law stress input: the rank class used to wait 2.5 ms for
    // them) -- and the one-wave classes alternating from 1, as they always did
    order[0] = 0;
    lane[0] = 0;
    order[1] = kDenseBin;
    lane[1] = 1;
    order[2] = kMidBin;
    lane[2] = 0;
    order[3] = kRankBin;
    lane[3] = 0;
    for (int k = 0; k < kWaveBins; k++) {
        order[4 + k] = light[k];
        lane[4 + k] = 3 + k;
    }
    static_assert(kNumBins == kWaveBins + 4, "every class has a position");
}

// the hub rows (class kDenseBin) of a multiply, largest first when there are few enough to rank
static void hub_order(bspgemm_context *ctx, int b, int n, const RowRec *&rec, const long long *&recpre, hipStream_t sx)
{
    if (b != kDenseBin || n < 2 || n > kHeavySortMax) return;
    launch_order_heavy(rec, recpre, n, ctx->hub_rec, ctx->hub_pre, sx);
    rec = ctx->hub_rec;
    recpre = ctx->hub_pre;
}

// BSPGEMM_OPT_CHECK: the device error word is cleared and B's derived tables are verified against its row_ptr before
// the prepass uses them; check_verdict (after the multiply's last synchronisation) turns a set bit into a failure
static bspgemm_status check_arm(bspgemm_context *ctx, const bspgemm_matrix *B, hipStream_t s)
{
    if (!ctx->check) return BSPGEMM_OK;
    HIPCHK(hipMemsetAsync(ctx->d_err, 0, sizeof(unsigned), s));
    launch_check_tables(B->d_row_ptr, B->rows, B->d_deg8, B->blk8_state == 1 ? B->d_blk8 : nullptr,
                        B->pad_state == 1 ? B->d_row_ptr_pad : nullptr, ctx->d_err, s);
    return BSPGEMM_OK;
}
static bspgemm_status check_verdict(bspgemm_context *ctx)
{
    if (!ctx->check || !ctx->h->err) return BSPGEMM_OK;
    if (ctx->h->err & kErrStaleTable)
        return FAIL(BSPGEMM_ERR_INVALID, "operand B was rewritten in place: its derived tables do not match its row_ptr (call bspgemm_matrix_invalidate)");
    return FAIL(BSPGEMM_ERR_INVALID, "a row gathered more products than its capacity class holds (operand changed during the multiply?)");
}

// closes the multiply's stat slot (its events have all completed: the caller has synchronised)
static void close_slot(bspgemm_context *ctx, int R, const HostScalars *h, long long products, long long nnz_c,
                       const int (*cls_n)[kNumBins], int mid_cap, int rank_cap = 0)
{
    bspgemm_context::StatSlot &sl = ctx->slots[ctx->slot_head];
    sl.R = R;
    sl.cls_timed = ctx->class_timing;
    sl.mid_cap = mid_cap;
    sl.rank_cap = rank_cap;
    sl.h = *h;
    sl.products = products;
    sl.nnz_c = nnz_c;
    memcpy(sl.cls_n, cls_n, sizeof sl.cls_n);
    sl.used = true;
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
    slot.flow = BSPGEMM_FLOW_EXACT;
    slot.class_streams = ctx->class_streams;
    slot.small = false;
    slot.checked = ctx->check;

    HIPCHK_B(hipEventRecord(slot.ev[0], s));
    HIPCHK_B(result_alloc(ctx, reinterpret_cast<void **>(&C->d_row_ptr), result_bytes_rowptr(R)));

    // ---- symbolic 1: per-row products, their prefix, capacity classes ---------------------
    const int scan_tiles = (R + 2047) / 2048;
    const int heavy_cols = B->cols > 0 ? B->cols : 1;
    // the extents ab[] come from the prepass, as in the other flow: both passes of the one-wave classes read them
    if (bspgemm_status st = ensure_pad(B)) return bail(st);        // (first use as B: the padded copy, then the blocked table over it)
    if (bspgemm_status st = ensure_blk8(B)) return bail(st);       // (wrapped device arrays: first use)
    if (bspgemm_status st = check_arm(ctx, B, s)) return bail(st);
    slot.prepass_kernel = B->blk8_state == 1 ? 1 : 0;
    slot.padded = B->pad_state == 1;
    const int *Bcol = B->gather_col();                     // B.col_idx, or its padded copy (the extents in ab[] point into it)
    launch_row_work(A->d_row_ptr, A->d_col_idx, B->d_row_ptr, B->blk8_state == 1 ? B->d_blk8 : nullptr,
                    B->pad_state == 1 ? B->d_row_ptr_pad : nullptr, B->pad_state == 1 ? B->d_ext : nullptr, row_begin, row_end,
                    ctx->F, ctx->ab, s);
    launch_scan_and_bin(ctx->F, R, row_begin, A->d_row_ptr, ctx->Fprefix, ctx->partials, ctx->bin_tiles,
                        ctx->bin_count, ctx->rec, ctx->recpre, ctx->cnt, heavy_cols, ctx->hpartials, mid_cap_for_cols(B->cols), rank_cap_for_cols(B->cols), 0, s);
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
    int order[kNumBins], lane_of[kNumBins] = {};
    if (R > 0) {
        class_order(h->bin_count, totalF, order, lane_of);
        HIPCHK_B(fork(ctx->ev_tile[0][0]));
        for (int pos = 1; pos < kNumBins; pos++) {
            const int b = order[pos];
            const int n = h->bin_count[b];
            cls_n[0][b] = n;
            if (n <= 0) continue;
            hipStream_t sx = lanes[lane_of[pos] % nlanes];
            const RowRec *rec = ctx->rec + bin_start[b];
            if (ctx->class_timing) HIPCHK_B(hipEventRecord(slot.ev_cls[0][b][0], sx));
            if (b <= kWaveBins) {
                // the numeric kernel without its emit half: |C_i| = F_i as soon as every product is seen to sit alone
                // in its 32-column slot, the level-0 masks are only built and counted for the other rows
                launch_wave_rows(b, wave_levels_for_cols(B->cols), ctx->ab, Bcol, B->cols, rec, nullptr, nullptr, n,
                                 row_begin, nullptr, ctx->cnt, ctx->d_err, sx, true);
            } else {
                const long long *hpre = ctx->recpre + bin_start[b];
                hub_order(ctx, b, n, rec, hpre, sx);
                HIPCHK_B(launch_dense_rows(b, ctx->ab, Bcol, B->nnz, B->cols, rec, hpre, n,
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
            hipStream_t sx = b > kWaveBins ? sC : lanes[lane_of[pos] % nlanes];
            if (ctx->class_timing) HIPCHK_B(hipEventRecord(slot.ev_cls[1][b][0], sx));
            if (b <= kWaveBins)
                launch_wave_rows(b, levels, ctx->ab, Bcol, B->cols, rec, recpre, C->d_row_ptr, n, row_begin,
                                 C->d_col_idx, nullptr, ctx->d_err, sx);
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
    if (ctx->check) HIPCHK_B(hipMemcpyAsync(&h->err, ctx->d_err, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipStreamSynchronize(s));
    (void)synced;
    if (bspgemm_status st = check_verdict(ctx)) return bail(st);
    C->nnz = h->nnzC;
    close_slot(ctx, R, h, totalF, C->nnz, cls_n, mid_cap_for_cols(B->cols), rank_cap_for_cols(B->cols));
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
    slot.flow = BSPGEMM_FLOW_UPPER_BOUND;
    slot.class_streams = ctx->class_streams;
    slot.small = false;
    slot.checked = ctx->check;

    HIPCHK_B(hipEventRecord(slot.ev[0], s));
    HIPCHK_B(result_alloc(ctx, reinterpret_cast<void **>(&C->d_row_ptr), result_bytes_rowptr(R)));
    if (bspgemm_status st = ensure_pad(B)) return bail(st);
    if (bspgemm_status st = ensure_blk8(B)) return bail(st);
    if (bspgemm_status st = check_arm(ctx, B, s)) return bail(st);
    slot.prepass_kernel = B->blk8_state == 1 ? 1 : 0;
    slot.padded = B->pad_state == 1;
    const int *Bcol = B->gather_col();                     // B.col_idx, or its padded copy (the extents in ab[] point into it)
    launch_row_work(A->d_row_ptr, A->d_col_idx, B->d_row_ptr, B->blk8_state == 1 ? B->d_blk8 : nullptr,
                    B->pad_state == 1 ? B->d_row_ptr_pad : nullptr, B->pad_state == 1 ? B->d_ext : nullptr, row_begin, row_end,
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
                        Fm ? 0 : rank_cap_for_cols(B->cols), B->cols > 0 ? B->cols : 1, s, ctx->d_prep, Fm ? ctx->F : nullptr);
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
        int lane_of[kNumBins];
        class_order(h->bin_count, h->products, order, lane_of);
        for (int pos = 1; pos < kNumBins; pos++) {
            const int b = order[pos];
            const int n = h->bin_count[b];
            cls_n[1][b] = n;
            if (n <= 0) continue;
            hipStream_t sx = lanes[lane_of[pos] % nlanes];
            const RowRec *rec = ctx->rec + bin_start[b];
            const long long *recpre = ctx->recpre + bin_start[b];
            if (ctx->class_timing) HIPCHK_B(hipEventRecord(slot.ev_cls[1][b][0], sx));
            if (!Fm) hub_order(ctx, b, n, rec, recpre, sx);
            if (!Fm && b <= kWaveBins)
                launch_wave_rows(b, levels, ctx->ab, Bcol, B->cols, rec, recpre, nullptr, n, row_begin,
                                 ctx->tmp, ctx->cnt, ctx->d_err, sx);
            else if (!Fm)
                HIPCHK_B(launch_dense_rows(b, ctx->ab, Bcol, B->nnz, B->cols, rec, recpre, n, row_begin, ctx->tmp,
                                           ctx->cnt, sx));
            else if (b <= kWaveBins && wave_masked_supported(B->cols))
                launch_wave_masked(b, ctx->ab, Bcol, B->cols, Fm->d_row_ptr, Fm->d_col_idx, rec, recpre, n,
                                   row_begin, ctx->tmp, ctx->cnt, sx);
            else
                HIPCHK_B(launch_dense_rows_masked(ctx->ab, Bcol, B->nnz, B->cols, rec, recpre, n, row_begin,
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
    if (ctx->check) HIPCHK_B(hipMemcpyAsync(&h->err, ctx->d_err, sizeof(unsigned), hipMemcpyDeviceToHost, s));
    HIPCHK_B(hipStreamSynchronize(s));
    if (bspgemm_status st = check_verdict(ctx)) return bail(st);
    C->nnz = h->nnzC;
    close_slot(ctx, R, h, R > 0 ? h->products : 0, C->nnz, cls_n, mid_cap_for_cols(B->cols), Fm ? 0 : rank_cap_for_cols(B->cols));
    *out = C;
    return BSPGEMM_OK;
}

// Small products (csrc/small.hip): five launches, no size goes to the host before the end, ONE read-back.  *bailed = true
// (status OK, no result) when the device found that the product does not fit the path: the caller runs the general flow.
static bool small_eligible(const bspgemm_context *ctx, const bspgemm_matrix *A, const bspgemm_matrix *B, int R)
{
    if (ctx->small == 0 || R <= 0 || R > kSmallMaxRows || A->nnz > kSmallMaxNnzA) return false;
    if (ctx->small == 1) return true;
    // automatic: expected products = A's nonzeros x B's mean row length must leave room below the path's capacity
    const double mean_b = B->rows > 0 ? (double)B->nnz / (double)B->rows : 0.0;
    return (double)A->nnz * mean_b <= 0.5 * kSmallMaxProducts;
}

static bspgemm_status multiply_small(bspgemm_context *ctx, const bspgemm_matrix *A, const bspgemm_matrix *B,
                                     int row_begin, int row_end, bspgemm_result **out, bool *bailed)
{
    *bailed = false;
    if (!out) return FAIL(BSPGEMM_ERR_INVALID, "result pointer is NULL");
    *out = nullptr;
    if (bspgemm_status st = check_operands(ctx, A, B, row_begin, row_end)) return st;
    if (bspgemm_status st = use_device(ctx)) return st;
    const int R = row_end - row_begin;
    hipStream_t s = ctx->stream;
    if (bspgemm_status st = ensure_rows(ctx, (size_t)R + 1)) return st;
    if (bspgemm_status st = ensure_tmp(ctx, (size_t)kSmallMaxProducts + 1)) return st;
    bspgemm_result *C = new (std::nothrow) bspgemm_result{ctx, R, 0, nullptr, nullptr, 0};
    if (!C) return FAIL(BSPGEMM_ERR_ALLOC, "result");
    auto bail = [&](bspgemm_status st) {
        hipStreamSynchronize(s);
        bspgemm_result_free(C);
        return st;
    };
    ctx->slot_head = (ctx->slot_head + 1) % bspgemm_context::kStatSlots;
    bspgemm_context::StatSlot &slot = ctx->slots[ctx->slot_head];
    slot.used = false;
    slot.flow = BSPGEMM_FLOW_UPPER_BOUND;                  // (rows are placed by their product count and squeezed together)
    slot.class_streams = 1;
    slot.small = true;
    slot.checked = false;
    slot.prepass_kernel = 2;
    HIPCHK_B(hipEventRecord(slot.ev[0], s));
    HIPCHK_B(result_alloc(ctx, reinterpret_cast<void **>(&C->d_row_ptr), result_bytes_rowptr(R)));
    HIPCHK_B(result_alloc(ctx, reinterpret_cast<void **>(&C->d_col_idx), result_bytes_colidx(kSmallMaxProducts)));
    C->col_cap = kSmallMaxProducts;
    launch_small(A->d_row_ptr, A->d_col_idx, B->d_row_ptr, B->d_col_idx, row_begin, R, ctx->F, ctx->Fprefix,
                 reinterpret_cast<int *>(ctx->rec), ctx->cnt, ctx->tmp, C->d_row_ptr, C->d_col_idx, ctx->d_small_tiles, ctx->d_small, s);
    HIPCHK_B(hipGetLastError());
    HostScalars *h = ctx->h;
    HIPCHK_B(hipMemcpyAsync(&h->small, ctx->d_small, sizeof(SmallScalars), hipMemcpyDeviceToHost, s));
    for (int e = 1; e <= 4; e++) HIPCHK_B(hipEventRecord(slot.ev[e], s));   // one phase: everything is "total"
    HIPCHK_B(hipStreamSynchronize(s));
    if (h->small.bail) {
        bspgemm_result_free(C);
        *bailed = true;
        return BSPGEMM_OK;
    }
    C->nnz = h->small.nnzC;
    h->totalF = h->small.totalF;
    h->nnzC = h->small.nnzC;
    h->a_lo = h->small.a_lo;
    h->a_hi = h->small.a_hi;
    memset(h->bin_count, 0, sizeof h->bin_count);
    h->bin_count[0] = R - h->small.nonempty;               // (one "class": every non-empty row is sorted by one wave)
    h->bin_count[kWaveBins] = h->small.nonempty;
    int cls_n[2][kNumBins] = {};
    close_slot(ctx, R, h, h->small.totalF, C->nnz, cls_n, mid_cap_for_cols(B->cols));
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
    if (A && B && ctx->flow != BSPGEMM_FLOW_EXACT && small_eligible(ctx, A, B, row_end - row_begin)) {
        bool bailed = false;
        const bspgemm_status st = multiply_small(ctx, A, B, row_begin, row_end, out, &bailed);
        if (st != BSPGEMM_OK || !bailed) return st;        // done (or failed); a product that did not fit falls through
    }
    if (ctx->flow == BSPGEMM_FLOW_EXACT) return multiply_exact(ctx, A, B, row_begin, row_end, out);
    bspgemm_status st = multiply_upper_bound(ctx, A, B, nullptr, row_begin, row_end, out);
    if (st == BSPGEMM_ERR_ALLOC && ctx->flow == BSPGEMM_FLOW_AUTO) {
        (void)hipGetLastError();
        st = multiply_exact(ctx, A, B, row_begin, row_end, out);
    }
    return st;
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
                        ctx->rec, ctx->recpre, ctx->cnt, 0, nullptr, mid_cap_for_cols(B->cols), rank_cap_for_cols(B->cols), 0, ctx->stream);
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

