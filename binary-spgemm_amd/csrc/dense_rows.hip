// dense_rows.hip -- heavy rows (F_i > 2048 products): one 1024-thread workgroup per A-row with
// a dense column bitmap held in LDS, swept over column windows of up to 2^20 bits (128 KiB).
//
// This is the literal GPU form of the reference's accumulator (final/SpGEMM_mpi_omp.c:21,38-42):
// xb[k] becomes bit k of the LDS bitmap, test-and-set becomes ds_or_b32, and the quickSort of
// the row (:47) disappears because the bitmap is read out in column order.  Rows this heavy have
// many duplicate products, so the result is dense enough that scanning the window pays.
// When cols > 2^20 the row's products are re-gathered once per window (B is L2/MALL resident
// for a hub row: its B rows were just read by the previous window).
// Also holds the compaction kernels that squeeze the upper-bound-placed rows into C.col_idx.
#include "kernels.hpp"
#include "wave.hpp"

namespace bsp {

constexpr int kDenseThreads = 1024;
constexpr int kDenseMaxWords = 16384;        // 64-bit words per window = 2^20 columns = 128 KiB

__global__ __launch_bounds__(kDenseThreads) void k_dense_rows(const int2 *__restrict__ ab,
                                                              const int *__restrict__ Bcol,
                                                              int cols, int wwords,
                                                              const RowRec *__restrict__ rec,
                                                              const long long *__restrict__ recpre,
                                                              int row_begin,
                                                              int *__restrict__ tmp,
                                                              int *__restrict__ cnt)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u64 *bm = reinterpret_cast<u64 *>(lds_raw);
    u32 *bm32 = reinterpret_cast<u32 *>(lds_raw);
    __shared__ int wtot[kDenseThreads / 64];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < wwords; t += kDenseThreads) bm[t] = 0ull;
    __syncthreads();

    const RowRec q = rec[blockIdx.x];
    const int i = q.row;
    const int a0 = q.a0, a1 = q.a0 + q.alen;
    int *out = tmp + recpre[blockIdx.x];
    const long long W = (long long)wwords * 64;
    const int nwin = (int)(((long long)cols + W - 1) / W);
    const int group = tid >> 4, sub = tid & 15;                 // 64 groups of 16 lanes
    const int wpt = (wwords + kDenseThreads - 1) / kDenseThreads;
    int total = 0;

    for (int win = 0; win < nwin; win++) {
        const long long lo = (long long)win * W;
        for (int jj = a0 + group; jj < a1; jj += kDenseThreads / 16) {
            const int2 e = ab[jj];
            const int bs = e.x, be = e.x + e.y;
            for (int k = bs + sub; k < be; k += 16) {
                const long long c = (long long)Bcol[k] - lo;
                if (c >= 0 && c < W) atomicOr(&bm32[c >> 5], 1u << (c & 31));
            }
        }
        __syncthreads();
        // each thread owns `wpt` consecutive words: count, block-scan, emit in column order
        const int w0 = tid * wpt;
        const int w1 = (w0 + wpt < wwords) ? w0 + wpt : wwords;
        int c = 0;
        for (int w = w0; w < w1; w++) c += __popcll(bm[w]);
        const int inc = wave_incl_scan(c);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int off = inc - c, btotal = 0;
        for (int k = 0; k < kDenseThreads / 64; k++) {
            const int t = wtot[k];
            if (k < wave) off += t;
            btotal += t;
        }
        int pos = total + off;
        for (int w = w0; w < w1; w++) {
            u64 m = bm[w];
            bm[w] = 0ull;
            const int base = (int)(lo + (long long)w * 64);
            while (m) {
                out[pos++] = base | (int)__builtin_ctzll(m);
                m &= m - 1ull;
            }
        }
        total += btotal;
        __syncthreads();
    }
    if (tid == 0) cnt[i - row_begin] = total;
}

hipError_t launch_dense_rows(const int2 *ab, const int *Bcol, int cols,
                             const RowRec *rec, const long long *recpre, int nrows, int row_begin,
                             int *tmp, int *cnt, hipStream_t s)
{
    if (nrows <= 0) return hipSuccess;
    long long words = ((long long)cols + 63) / 64;
    if (words > kDenseMaxWords) words = kDenseMaxWords;
    if (words < 1) words = 1;
    const int bytes = (int)words * 8;
    static int attr_set_for = 0;
    if (bytes > attr_set_for) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_rows),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
        if (e != hipSuccess) return e;
        attr_set_for = 128 * 1024;
    }
    hipLaunchKernelGGL(k_dense_rows, dim3(nrows), dim3(kDenseThreads), bytes, s, ab, Bcol,
                       cols, (int)words, rec, recpre, row_begin, tmp, cnt);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------
// Compaction: every row was written at its upper-bound offset Fprefix[r]; now that the counts
// are scanned into C.row_ptr the rows are copied to their final place.  Pure streaming copy
// (4 B read + 4 B written per output nonzero), one wave per 8 consecutive rows; rows longer
// than 8192 entries are left to a workgroup-per-row kernel.
constexpr int kCompactBigRow = 8192;
constexpr int kCompactRows = 8;          // consecutive rows copied by one wave

__global__ __launch_bounds__(256) void k_compact(const int *__restrict__ tmp,
                                                 const long long *__restrict__ Fprefix,
                                                 const long long *__restrict__ row_ptr, int nrows,
                                                 int *__restrict__ col_idx)
{
    const int lane = threadIdx.x & 63;
    const long long wave_global = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    const long long r0 = wave_global * kCompactRows;
    if (r0 >= nrows) return;
    // lanes 0..8 fetch the row_ptr window, lanes 0..7 the source offsets
    long long rp = 0, fp = 0;
    if (lane <= kCompactRows && r0 + lane <= nrows) rp = row_ptr[r0 + lane];
    if (lane < kCompactRows && r0 + lane < nrows) fp = Fprefix[r0 + lane];
    for (int k = 0; k < kCompactRows; k++) {
        if (r0 + k >= nrows) break;
        const long long d0 = __shfl(rp, k, 64), d1 = __shfl(rp, k + 1, 64);
        const long long s0 = __shfl(fp, k, 64);
        const long long len = d1 - d0;
        if (len > kCompactBigRow) continue;
        for (long long t = lane; t < len; t += 64) col_idx[d0 + t] = tmp[s0 + t];
    }
}

__global__ __launch_bounds__(1024) void k_compact_big(const int *__restrict__ tmp,
                                                      const long long *__restrict__ Fprefix,
                                                      const long long *__restrict__ row_ptr,
                                                      const RowRec *__restrict__ rec, int row_begin,
                                                      int *__restrict__ col_idx)
{
    const int r = rec[blockIdx.x].row - row_begin;
    const long long d0 = row_ptr[r], len = row_ptr[r + 1] - d0, s0 = Fprefix[r];
    if (len <= kCompactBigRow) return;
    for (long long t = threadIdx.x; t < len; t += 1024) col_idx[d0 + t] = tmp[s0 + t];
}

void launch_compact(const int *tmp, const long long *Fprefix, const long long *row_ptr, int nrows,
                    int *col_idx, hipStream_t s)
{
    if (nrows <= 0) return;
    const long long rows_per_wg = 4ll * kCompactRows;
    const int grid = (int)((nrows + rows_per_wg - 1) / rows_per_wg);
    hipLaunchKernelGGL(k_compact, dim3(grid), dim3(256), 0, s, tmp, Fprefix, row_ptr, nrows, col_idx);
}

void launch_compact_big(const int *tmp, const long long *Fprefix, const long long *row_ptr,
                        const RowRec *rec, int nrows, int row_begin, int *col_idx, hipStream_t s)
{
    if (nrows <= 0) return;
    hipLaunchKernelGGL(k_compact_big, dim3(nrows), dim3(1024), 0, s, tmp, Fprefix, row_ptr, rec, row_begin,
                       col_idx);
}

}  // namespace bsp
