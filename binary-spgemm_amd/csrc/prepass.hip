// prepass.hip -- symbolic side of bspgemm_multiply: per-row work, scans, capacity binning.
//
// The reference has no counterpart: SpGEMM_bigslice (final/SpGEMM_mpi_omp.c:15-58) discovers
// a row's size while it fills it and grows Ccol with realloc (:28-31).  On the GPU every row is
// sized up front by its product count F_i = sum_{j in A_i} |B_j| (an upper bound of |C_i|), which
// (a) fixes where each row's result lands without any inter-wave dependency and (b) selects
// the accumulator capacity class.  All kernels here are HBM-bound streaming/gather passes:
// row_work reads 4*nnzA (A.col_idx) + 8*nnzA (B.row_ptr pairs) bytes, the scans touch 8-16 B/row.
#include "kernels.hpp"
#include "wave.hpp"

namespace bsp {

// ---------------------------------------------------------------------------------------
// 8 lanes per A row: lanes stride the row's col_idx (coalesced 32-B pieces), gather the
// B.row_ptr pair of every nonzero, reduce over the 8 lanes.
__global__ __launch_bounds__(256) void k_row_work(const int *__restrict__ Arow,
                                                  const int *__restrict__ Acol,
                                                  const int *__restrict__ Brow,
                                                  int row_begin, int nrows,
                                                  long long *__restrict__ F)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(gid >> 3);
    const int sub = (int)(gid & 7);
    long long sum = 0;
    if (r < nrows) {
        const int a0 = Arow[row_begin + r], a1 = Arow[row_begin + r + 1];
        for (int jj = a0 + sub; jj < a1; jj += 8) {
            const int j = Acol[jj];
            sum += (long long)(Brow[j + 1] - Brow[j]);
        }
    }
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 4, 64);
    if (r < nrows && sub == 0) F[r] = sum;
}

void launch_row_work(const int *Arow, const int *Acol, const int *Brow,
                     int row_begin, int row_end, long long *F, hipStream_t s)
{
    const int nrows = row_end - row_begin;
    if (nrows <= 0) return;
    const long long threads = (long long)nrows * 8;
    const int grid = (int)((threads + 255) / 256);
    hipLaunchKernelGGL(k_row_work, dim3(grid), dim3(256), 0, s, Arow, Acol, Brow, row_begin, nrows, F);
}

// ---------------------------------------------------------------------------------------
// Three-kernel exclusive scan: block sums -> scan of the block sums -> apply.
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;   // 2048 values per workgroup

template <typename T>
__device__ __forceinline__ long long block_sum(long long v, long long *lds /* >= 4 */)
{
    v = wave_incl_scan64(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) lds[w] = v;
    __syncthreads();
    long long t = 0;
    for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += lds[i];
    __syncthreads();
    return t;
}

template <typename T>
__global__ __launch_bounds__(kScanThreads) void k_tile_sums(const T *__restrict__ in, int n,
                                                            long long *__restrict__ partials)
{
    __shared__ long long lds[4];
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    long long v = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++)
        if (base + k < n) v += (long long)in[base + k];
    const long long t = block_sum<T>(v, lds);
    if (threadIdx.x == 0) partials[blockIdx.x] = t;
}

// single workgroup: in-place exclusive scan of `m` tile sums; partials[m] = grand total
__global__ __launch_bounds__(1024) void k_scan_partials(long long *__restrict__ partials, int m)
{
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int base = 0; base < m; base += 1024) {
        const int i = base + threadIdx.x;
        const long long v = i < m ? partials[i] : 0;
        long long inc = wave_incl_scan64(v);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        long long woff = 0;
        for (int k = 0; k < w; k++) woff += wsum[k];
        long long total = 0;
        for (int k = 0; k < 16; k++) total += wsum[k];
        const long long carry = carry_s;
        if (i < m) partials[i] = carry + woff + inc - v;
        __syncthreads();
        if (threadIdx.x == 0) carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[m] = carry_s;
}

// capacity class of a row with F products: 0 empty, 1..6 one wave (64<<(b-1)), 7 dense window
__device__ __forceinline__ int bin_of(long long F)
{
    if (F <= 0) return 0;
    if (F > kMaxWaveCap) return 7;
    const int f = (int)F;
    // smallest b >= 1 with f <= 64 << (b-1)
    const int hb = 32 - __clz(f - 1);          // bits needed for f-1 (0 for f == 1)
    return hb <= 6 ? 1 : hb - 5;
}

template <typename T, bool BIN>
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(const T *__restrict__ in, int n,
                                                             const long long *__restrict__ partials,
                                                             long long *__restrict__ out,
                                                             int row_begin, int *__restrict__ bin_rows,
                                                             int *__restrict__ bin_count,
                                                             int *__restrict__ cnt)
{
    __shared__ long long wsum[4];
    __shared__ int lcount[kNumBins];
    __shared__ int lbase[kNumBins];
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    long long v[kScanItems];
    long long tsum = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        v[k] = (base + k < n) ? (long long)in[base + k] : 0;
        tsum += v[k];
    }
    if (BIN) {
        if (threadIdx.x < kNumBins) lcount[threadIdx.x] = 0;
    }
    const long long inc = wave_incl_scan64(tsum);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    long long off = partials[blockIdx.x] + inc - tsum;
    for (int k = 0; k < w; k++) off += wsum[k];
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        if (base + k < n) out[base + k] = off;
        off += v[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) out[n] = partials[gridDim.x];

    if (BIN) {
        // workgroup-aggregated append: one global atomic per (workgroup, bin)
        int myb[kScanItems], mypos[kScanItems];
#pragma unroll
        for (int k = 0; k < kScanItems; k++) {
            myb[k] = -1;
            if (base + k < n) {
                myb[k] = bin_of(v[k]);
                mypos[k] = atomicAdd(&lcount[myb[k]], 1);
                if (myb[k] == 0) cnt[base + k] = 0;
            }
        }
        __syncthreads();
        if (threadIdx.x < kNumBins) {
            const int c = lcount[threadIdx.x];
            lbase[threadIdx.x] = c ? atomicAdd(&bin_count[threadIdx.x], c) : 0;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kScanItems; k++)
            if (myb[k] > 0)
                bin_rows[(long long)myb[k] * n + lbase[myb[k]] + mypos[k]] = row_begin + base + k;
    }
}

template <typename T, bool BIN>
static void scan_impl(const T *in, int n, long long *prefix, long long *partials, int row_begin,
                      int *bin_rows, int *bin_count, int *cnt, hipStream_t s)
{
    if (n <= 0) {
        hipMemsetAsync(prefix, 0, sizeof(long long), s);
        return;
    }
    const int tiles = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL((k_tile_sums<T>), dim3(tiles), dim3(kScanThreads), 0, s, in, n, partials);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, partials, tiles);
    hipLaunchKernelGGL((k_scan_apply<T, BIN>), dim3(tiles), dim3(kScanThreads), 0, s, in, n, partials,
                       prefix, row_begin, bin_rows, bin_count, cnt);
}

void launch_scan_and_bin(const long long *F, int n, int row_begin, long long *prefix,
                         long long *partials, int *bin_rows, int *bin_count, int *cnt, hipStream_t s)
{
    hipMemsetAsync(bin_count, 0, kNumBins * sizeof(int), s);
    scan_impl<long long, true>(F, n, prefix, partials, row_begin, bin_rows, bin_count, cnt, s);
}

void launch_scan_counts(const int *cnt, int n, long long *prefix, long long *partials, hipStream_t s)
{
    scan_impl<int, false>(cnt, n, prefix, partials, 0, nullptr, nullptr, nullptr, s);
}

// ---------------------------------------------------------------------------------------
__global__ void k_rebase_i32(int *p, int n, int base)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] -= base;
}
void launch_rebase_i32(int *row_ptr, int n, int base, hipStream_t s)
{
    if (n <= 0 || base == 0) return;
    hipLaunchKernelGGL(k_rebase_i32, dim3((n + 255) / 256), dim3(256), 0, s, row_ptr, n, base);
}

__global__ void k_add_base_i64(long long *dst, const long long *src, int n, long long base)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = src[i] + base;
}
void launch_add_base_i64(long long *dst, const long long *src, int n, long long base, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_add_base_i64, dim3((n + 255) / 256), dim3(256), 0, s, dst, src, n, base);
}

}  // namespace bsp
