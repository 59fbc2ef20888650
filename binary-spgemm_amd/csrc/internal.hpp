// internal.hpp -- what the translation units behind include/bspgemm.h share: the error macros, the handle
// structs, the workspace / result-cache helpers (context.hip) that the flows (multiply.hip), the int32 drop-ins
// (dropin.hip) and the communicator layer (comm.hip) use.  Not installed; nothing here is part of the C ABI.
#pragma once
#include "../../include/bspgemm.h"
#include "kernels.hpp"

#include <hip/hip_runtime.h>

#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <chrono>
#include <new>

// ------------------------------------------------------------------ errors ---------------
// text of the last failure on this thread (bspgemm_last_error); defined in context.hip
extern thread_local char bspgemm_err_text[512];
#define g_err bspgemm_err_text

static inline bspgemm_status fail(bspgemm_status st, const char *what, const char *file, int line)
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

// ------------------------------------------------------------------ objects --------------
using bsp::kNumBins;
using bsp::PrepScalars;
using bsp::RowRec;
struct HostScalars {
    long long totalF;
    long long nnzC;
    int bin_count[kNumBins];
    int a_lo, a_hi;
    long long products;                 // true product count of a masked multiply (totalF is the mask total there)
    long long heavy_total;              // entries the heavy rows may need in the workspace
    PrepScalars prep;                   // upper-bound flow: the prepass results, fetched in one copy
    unsigned err;                       // ctx->d_err, read back when BSPGEMM_OPT_CHECK is on
    bsp::SmallScalars small;            // small path: its one read-back
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
        int mid_cap = 0, rank_cap = 0;
        HostScalars h = {};
        long long products = 0, nnz_c = 0;
        int cls_n[2][kNumBins] = {};
        bool used = false;
        int flow = 0, prepass_kernel = 0, class_streams = 0;   // which path ran (bspgemm_stats)
        bool small = false, checked = false, padded = false;
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
    PrepScalars *d_prep = nullptr;      // device side of HostScalars::prep
    bsp::SmallScalars *d_small = nullptr; // device side of HostScalars::small
    bsp::SmallTiles *d_small_tiles = nullptr;   // scan scratch of the small path
    unsigned *d_err = nullptr;          // device error word of the accumulate kernels (kErrCapacity | kErrStaleTable)
    int *chunk_row = nullptr;           // compaction: row of every kCompactGran-th output (left by the count scan)
    size_t chunk_cap = 0;
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
    int pad_rows = 0;                   // BSPGEMM_PAD_ROWS / BSPGEMM_OPT_PADDED_ROWS: 0 never (default), 1 always, -1 per operand: gather from a padded copy of B.col_idx
    int small = -1;                     // BSPGEMM_SMALL / BSPGEMM_OPT_SMALL_PATH: -1 automatic, 0 never, 1 whenever the product fits
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
    // padded copy of col_idx: every row on a 64-byte boundary (padded to a multiple of 16 entries), so that a gathered B row
    // touches ceil(len / 16) 64-byte sectors instead of one more; built on first use as B (ensure_pad) when it pays
    mutable int *d_col_pad = nullptr;
    mutable int *d_row_ptr_pad = nullptr;  // rows + 1: where row j starts in d_col_pad
    mutable int2 *d_ext = nullptr;         // {start in d_col_pad, length} per row: gathered by k_row_work when there is no blocked table
    mutable int pad_state = 0;             // 0 undecided, 1 in use, 2 not for this operand
    // what the accumulate kernels gather from: the padded copy when it exists
    const int *gather_col() const { return pad_state == 1 ? d_col_pad : d_col_idx; }
};

bspgemm_status ensure_deg8(const bspgemm_matrix *m);
bspgemm_status ensure_blk8(const bspgemm_matrix *m);
bspgemm_status ensure_pad(const bspgemm_matrix *m);   // before ensure_blk8: the blocked table carries padded bases

struct bspgemm_result {
    bspgemm_context *ctx;
    int rows;
    long long nnz;
    long long *d_row_ptr;
    int *d_col_idx;
    long long col_cap;      // entries allocated for d_col_idx (upper bound F >= nnz)
};

bspgemm_status use_device(bspgemm_context *ctx);

// ------------------------------------------------------------------ workspace (context.hip) ---
bspgemm_status ensure_rows(bspgemm_context *ctx, size_t rows);
bspgemm_status ensure_ab(bspgemm_context *ctx, size_t pairs);
bspgemm_status ensure_tmp(bspgemm_context *ctx, size_t ints);
bspgemm_status ensure_chunk_rows(bspgemm_context *ctx, size_t entries);
// result buffers: best fit from the context's cache of freed results, else hipMalloc
bool result_cached(const bspgemm_context *ctx, size_t bytes);
hipError_t result_alloc(bspgemm_context *ctx, void **out, size_t bytes);
void result_release(bspgemm_context *ctx, void *p, size_t bytes);
static inline size_t result_bytes_rowptr(int rows) { return ((size_t)rows + 1) * sizeof(long long); }
static inline size_t result_bytes_colidx(long long nnz) { return ((size_t)nnz + 4) * sizeof(int); }

// one line on stderr for a failed drop-in ("SpGEMM_hip: <status>: <last error>"), returns the status as int (dropin.hip)
int dropin_fail(const char *fn, bspgemm_status st);
