// dropin.hip -- the int32 drop-ins with the reference's own argument lists and ownership rules:
// SpGEMM_hip <-> SpGEMM_omp (final/SpGEMM_mpi_omp.c:71-74), SpGEMM_hip_bigslice <-> SpGEMM_bigslice (:15-18),
// SpGEMM_hip_mat <-> SpGEMM_mat (Matlab/inc/BSpGEMM.h:2-4), SpGEMM_hip_masked <-> SpGEMM_masked (:232-235).
// Host arrays in, malloc'ed / caller-owned host arrays out; everything in between is the native handle API.
#include "internal.hpp"

#include <sys/mman.h>
#include <unistd.h>
#include <mutex>
#include <thread>
#include <atomic>
#include <vector>

using namespace bsp;

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

int dropin_fail(const char *fn, bspgemm_status st)
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
    // The early destination is sized by the BOUND, which heavy merging can leave far above nnz(C): it is only used while it
    // stays below half of the host memory that is free right now (and below INT_MAX entries); otherwise the result is
    // malloc'ed with its true size after the multiply, as before.
    bool early_fits = bound >= 0 && bound <= INT_MAX;
    if (early_fits) {
        const long pages = sysconf(_SC_AVPHYS_PAGES), psize = sysconf(_SC_PAGESIZE);
        if (pages > 0 && psize > 0 && (double)bound * sizeof(int) > 0.5 * (double)pages * (double)psize) early_fits = false;
    }
    if (early_fits) {
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
        // a copy that fails after others were queued must not leave a DMA writing into memory that is about to be
        // unpinned and freed: every exit drains the stream first
        auto run = [&]() -> bspgemm_status {
            HIPCHK(hipMemcpyAsync(rp64, C->d_row_ptr, ((size_t)rows + 1) * sizeof(long long), hipMemcpyDeviceToHost, s));
            for (size_t k = 0, off = 0; off < need; k++, off += kPiece) {
                while (pieces_ready.load(std::memory_order_acquire) <= (long long)k) {
                    if (early_failed) return FAIL(BSPGEMM_ERR_ALLOC, "early destination");
                    usleep(50);
                }
                const size_t len = (need - off < kPiece) ? need - off : kPiece;
                HIPCHK(hipMemcpyAsync(reinterpret_cast<char *>(early) + off, reinterpret_cast<const char *>(C->d_col_idx) + off, len,
                                      hipMemcpyDeviceToHost, s));
            }
            return BSPGEMM_OK;
        };
        const bspgemm_status st = run();
        early_stop = true;                                 // pieces beyond nnz(C) are not needed
        const hipError_t e = hipStreamSynchronize(s);
        if (st) return st;
        HIPCHK(e);
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

