// small.hip -- the single-round-trip path for SMALL products (round 4).
//
// The general flows cost a product of any size ~25 kernel launches and two host round trips (the class counts must
// reach the host before the class kernels can be launched): 0.15 ms for the reference's own validity fixture
// (Matlab/validity_test.mtx: 12 502 products), which one CPU core multiplies in 0.79 ms -- launch latency,
// not work.  A product with few A-nonzeros takes this path instead: FIVE launches, no size ever goes to the host before
// the end, one read-back.
//     k_small_sizes   F_i = sum of |B_j| over A_i (8 lanes per row, B.row_ptr pairs read directly) + sums per 32 rows
//     k_small_plan    256 rows per workgroup: every workgroup adds up the 32-row sums before it (at most 4096: no
//                     second-level scan, no look-back), scans its rows: Fprefix, the list of the non-empty rows, totals;
//                     decides whether the product FITS this path (F <= kSmallMaxProducts, every F_i <= kSmallMaxRow)
//     k_small_rows    a wave takes 64 listed rows at a time: rows of at most kSmallTiny products are done by ONE LANE each
//                     (its products into 16 private LDS slots, insertion sort, duplicates dropped: the 64 rows' dependent
//                     loads -- A.row_ptr, A.col_idx, B.row_ptr, B.col_idx -- travel together), larger ones by the whole
//                     wave (products gathered into LDS, bitonic sort) -- the reference's own gather / sort / emit of one
//                     row (final/SpGEMM_mpi_omp.c:33-47) without the flag array; rows land at their upper-bound places
//     k_small_scan    256 rows per workgroup, the same way over the per-256-row sums k_small_rows left: C.row_ptr, nnz(C)
//     k_small_copy    rows squeezed into C.col_idx (64 listed rows per wave, short rows lane by lane)
// (The first version scanned with ONE workgroup: 112 + 85 us of its 228 us on the validity fixture were those two kernels.)
// If the product does not fit, every kernel after the plan is a no-op, the host sees `bail` in the one read-back and
// runs the general flow (one wasted round trip).  Same CSR, bit for bit, as the other flows (tests/test_gpu_parity.py).
#include "kernels.hpp"
#include "wave.hpp"

namespace bsp {

constexpr int kSmallTiny = 16;               // products of a row that one lane handles alone

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_small_sizes(const int *__restrict__ Arow, const int *__restrict__ Acol,
                                                     const int *__restrict__ Brow, int row_begin, int nrows,
                                                     long long *__restrict__ F, SmallTiles *__restrict__ tl)
{
    __shared__ long long s_f[4];
    __shared__ int s_nz[4], s_mx[4];
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int r = gid >> 3, sub = gid & 7;
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
    // this workgroup's 32 rows: sum, number of non-empty rows, largest row
    long long f = (sub == 0 && r < nrows) ? sum : 0;
    int nz = f > 0, mx = f > 0x7fffffffll ? 0x7fffffff : (int)f;
#pragma unroll
    for (int d = 8; d < 64; d <<= 1) {
        f += __shfl_xor(f, d, 64);
        nz += __shfl_xor(nz, d, 64);
        mx = max(mx, __shfl_xor(mx, d, 64));
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { s_f[w] = f; s_nz[w] = nz; s_mx[w] = mx; }
    __syncthreads();
    if (threadIdx.x == 0) {
        tl->f32[blockIdx.x] = s_f[0] + s_f[1] + s_f[2] + s_f[3];
        tl->nz32[blockIdx.x] = s_nz[0] + s_nz[1] + s_nz[2] + s_nz[3];
        tl->mx32[blockIdx.x] = max(max(s_mx[0], s_mx[1]), max(s_mx[2], s_mx[3]));
    }
}

// block-wide (256 threads) exclusive scan of one value per thread; also the block total
template <typename T>
__device__ __forceinline__ T block_excl_256(T mine, T *s_w, T *total)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T inc = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const T y = __shfl_up(inc, d, 64);
        if (lane >= d) inc += y;
    }
    if (lane == 63) s_w[w] = inc;
    __syncthreads();
    T off = inc - mine, tot = 0;
    for (int k = 0; k < 4; k++) {
        if (k < w) off += s_w[k];
        tot += s_w[k];
    }
    *total = tot;
    __syncthreads();
    return off;
}
// block-wide sum / max of one value per thread (256 threads)
template <typename T>
__device__ __forceinline__ T block_sum_256(T v, T *s_w)
{
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = v;
    __syncthreads();
    const T r = s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
    return r;
}

// workgroup b owns the rows [256 b, 256 b + 256) = the 32-row sums [8 b, 8 b + 8) of k_small_sizes
__global__ __launch_bounds__(256) void k_small_plan(const long long *__restrict__ F, int nrows, const int *__restrict__ Arow,
                                                    int row_begin, long long *__restrict__ Fprefix, int *__restrict__ list,
                                                    int *__restrict__ cnt, SmallTiles *__restrict__ tl, SmallScalars *__restrict__ sc)
{
    __shared__ long long s_ll[4];
    __shared__ int s_i[4];
    const int t = threadIdx.x;
    const int ntile32 = (nrows + 31) / 32;
    // everything before this workgroup, and the totals: each thread adds its share of the 32-row sums
    long long fb = 0, ft = 0;
    int nb = 0, nt = 0, mx = 0;
    for (int k = t; k < ntile32; k += 256) {
        const long long f = tl->f32[k];
        const int z = tl->nz32[k];
        ft += f;
        nt += z;
        mx = max(mx, tl->mx32[k]);
        if (k < 8 * (int)blockIdx.x) { fb += f; nb += z; }
    }
    const long long total = block_sum_256<long long>(ft, s_ll);
    const long long before = block_sum_256<long long>(fb, s_ll);
    const int ntotal = block_sum_256<int>(nt, s_i);
    const int nbefore = block_sum_256<int>(nb, s_i);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) mx = max(mx, __shfl_xor(mx, d, 64));
    if ((t & 63) == 0) s_i[t >> 6] = mx;
    __syncthreads();
    const int maxf = max(max(s_i[0], s_i[1]), max(s_i[2], s_i[3]));
    __syncthreads();
    const bool fits = total <= kSmallMaxProducts && maxf <= kSmallMaxRow;
    const int r = blockIdx.x * 256 + t;
    const long long f = r < nrows ? F[r] : 0;
    long long tot_ll;
    int tot_i;
    const long long off = before + block_excl_256<long long>(f, s_ll, &tot_ll);
    const int pos = nbefore + block_excl_256<int>(f > 0 ? 1 : 0, s_i, &tot_i);
    if (r < nrows) {
        Fprefix[r] = off;
        cnt[r] = 0;
        if (f > 0 && fits) list[pos] = r;
    }
    if (t == 0) tl->c256[blockIdx.x] = 0;                          // k_small_rows adds the row sizes up here
    if (blockIdx.x == 0 && t == 0) {
        Fprefix[nrows] = total;
        sc->totalF = total;
        sc->nonempty = ntotal;
        sc->bail = fits ? 0 : 1;
        sc->nnzC = 0;
        sc->a_lo = Arow[row_begin];
        sc->a_hi = Arow[row_begin + nrows];
    }
}

// ---------------------------------------------------------------------------------------
constexpr int kSmallWaves = 4;
constexpr int kTinyStride = kSmallTiny + 1;  // odd lane stride: the lanes' private slots fall into different banks

__global__ __launch_bounds__(64 * kSmallWaves) void k_small_rows(const int *__restrict__ Arow, const int *__restrict__ Acol,
                                                                 const int *__restrict__ Brow, const int *__restrict__ Bcol,
                                                                 int row_begin, const int *__restrict__ list,
                                                                 const long long *__restrict__ Fprefix,
                                                                 const SmallScalars *__restrict__ sc,
                                                                 int *__restrict__ tmp, int *__restrict__ cnt, SmallTiles *__restrict__ tl)
{
    __shared__ u32 s_keys[kSmallWaves][kSmallMaxRow];
    __shared__ u32 s_tiny[kSmallWaves][64 * kTinyStride];
    if (sc->bail) return;                                          // (uniform: written by the kernel before this one)
    const int nlist = sc->nonempty;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    u32 *keys = s_keys[wv];
    u32 *mykeys = s_tiny[wv] + lane * kTinyStride;
    const int nwaves = gridDim.x * kSmallWaves;
    for (int k0 = (blockIdx.x * kSmallWaves + wv) * 64; k0 < nlist; k0 += nwaves * 64) {
        // ---- 64 rows, one per lane ---------------------------------------------------------------
        int r = -1, a0 = 0, a1 = 0, F = 0;
        long long pre = 0;
        if (k0 + lane < nlist) {
            r = list[k0 + lane];
            a0 = Arow[row_begin + r];
            a1 = Arow[row_begin + r + 1];
            pre = Fprefix[r];
            F = (int)(Fprefix[r + 1] - pre);
        }
        if (r >= 0 && F <= kSmallTiny) {
            // the lane's own row: gather (at most 16 products), insertion sort, squeeze, write
            int n = 0;
            for (int s = a0; s < a1; s++) {
                const int j = Acol[s];
                const int b0 = Brow[j], b1 = Brow[j + 1];
                for (int t = b0; t < b1 && n < kSmallTiny; t++) mykeys[n++] = (u32)Bcol[t];
            }
            for (int i = 1; i < n; i++) {
                const u32 v = mykeys[i];
                int p = i - 1;
                while (p >= 0 && mykeys[p] > v) { mykeys[p + 1] = mykeys[p]; p--; }
                mykeys[p + 1] = v;
            }
            int *out = tmp + pre;
            int written = 0;
            for (int i = 0; i < n; i++)
                if (i == 0 || mykeys[i] != mykeys[i - 1]) out[written++] = (int)mykeys[i];
            cnt[r] = written;
            atomicAdd(&tl->c256[r >> 8], written);
        }
        // ---- the larger rows of the batch, one after the other, by the whole wave -----------------
        u64 bigrows = __ballot(r >= 0 && F > kSmallTiny);
        while (bigrows) {
            const int src = (int)__builtin_ctzll(bigrows);
            bigrows &= bigrows - 1ull;
            const int rr = wave_bcast(r, src), ra0 = wave_bcast(a0, src), ra1 = wave_bcast(a1, src), rF = wave_bcast(F, src);
            const long long rpre = (long long)(((u64)(u32)wave_bcast((int)(u32)((u64)pre >> 32), src) << 32) | (u32)wave_bcast((int)(u32)pre, src));
            int N = 64;
            while (N < rF) N <<= 1;
            for (int i = lane; i < N; i += 64) keys[i] = ~0u;
            wave_lds_fence();
            // gather: 64 sources at a time, lane s copies B row s to its place in product order
            int done = 0;
            for (int s0 = ra0; s0 < ra1; s0 += 64) {
                int b0 = 0, len = 0;
                if (s0 + lane < ra1) {
                    const int j = Acol[s0 + lane];
                    b0 = Brow[j];
                    len = Brow[j + 1] - b0;
                }
                const int inc = wave_incl_scan(len);
                const int at = done + inc - len;
                int longest = len;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) longest = max(longest, __shfl_xor(longest, d, 64));
                for (int t = 0; t < longest; t++)
                    if (t < len && at + t < kSmallMaxRow) keys[at + t] = (u32)Bcol[b0 + t];
                done += wave_bcast(inc, 63);
            }
            wave_lds_fence();
            // bitonic sort of N keys (ascending); lane handles the pairs p = lane, lane + 64, ... of each stage
            for (int size = 2; size <= N; size <<= 1)
                for (int stride = size >> 1; stride > 0; stride >>= 1) {
                    for (int p = lane; p < (N >> 1); p += 64) {
                        const int lo = ((p & ~(stride - 1)) << 1) | (p & (stride - 1));
                        const int hi = lo | stride;
                        const bool up = (lo & size) == 0;
                        const u32 x = keys[lo], y = keys[hi];
                        if ((x > y) == up) { keys[lo] = y; keys[hi] = x; }
                    }
                    wave_lds_fence();
                }
            // emit the distinct keys in order
            int *out = tmp + rpre;
            int written = 0;
            for (int i0 = 0; i0 < rF; i0 += 64) {
                const int i = i0 + lane;
                const bool keep = i < rF && (i == 0 || keys[i] != keys[i - 1]);
                const u64 bal = __ballot(keep);
                if (keep) out[written + __popcll(bal & mask_lt(lane))] = (int)keys[i];
                written += __popcll(bal);
            }
            if (lane == 0) {
                cnt[rr] = written;
                atomicAdd(&tl->c256[rr >> 8], written);
            }
            wave_lds_fence();
        }
    }
}

// C.row_ptr = scan of cnt, nnz(C): workgroup b owns the rows [256 b, 256 b + 256) and adds up the 256-row sums before it
__global__ __launch_bounds__(256) void k_small_scan(const int *__restrict__ cnt, int nrows, long long *__restrict__ row_ptr,
                                                    const SmallTiles *__restrict__ tl, SmallScalars *__restrict__ sc)
{
    __shared__ long long s_ll[4];
    if (sc->bail) return;
    const int t = threadIdx.x;
    const int ntile = (nrows + 255) / 256;
    long long cb = 0, ct = 0;
    for (int k = t; k < ntile; k += 256) {
        const long long c = tl->c256[k];
        ct += c;
        if (k < (int)blockIdx.x) cb += c;
    }
    const long long total = block_sum_256<long long>(ct, s_ll);
    const long long before = block_sum_256<long long>(cb, s_ll);
    const int r = blockIdx.x * 256 + t;
    const long long c = r < nrows ? cnt[r] : 0;
    long long tot;
    const long long off = before + block_excl_256<long long>(c, s_ll, &tot);
    if (r < nrows) row_ptr[r] = off;
    if (blockIdx.x == 0 && t == 0) {
        row_ptr[nrows] = total;
        sc->nnzC = total;
    }
}

// rows from their upper-bound places to their final ones: 64 listed rows per wave; a short row is copied by its lane,
// the others by the whole wave
__global__ __launch_bounds__(256) void k_small_copy(const int *__restrict__ cnt, const long long *__restrict__ Fprefix,
                                                    const int *__restrict__ list, const int *__restrict__ tmp,
                                                    const long long *__restrict__ row_ptr, int *__restrict__ col_idx,
                                                    const SmallScalars *__restrict__ sc)
{
    if (sc->bail) return;
    const int nlist = sc->nonempty;
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x * 4;
    for (int k0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 64; k0 < nlist; k0 += nwaves * 64) {
        int n = 0;
        long long s = 0, d = 0;
        if (k0 + lane < nlist) {
            const int r = list[k0 + lane];
            n = cnt[r];
            s = Fprefix[r];
            d = row_ptr[r];
        }
        if (n <= kSmallTiny)
            for (int i = 0; i < n; i++) col_idx[d + i] = tmp[s + i];
        u64 bigrows = __ballot(n > kSmallTiny);
        while (bigrows) {
            const int src = (int)__builtin_ctzll(bigrows);
            bigrows &= bigrows - 1ull;
            const int rn = wave_bcast(n, src);
            const long long rs = (long long)(((u64)(u32)wave_bcast((int)(u32)((u64)s >> 32), src) << 32) | (u32)wave_bcast((int)(u32)s, src));
            const long long rd = (long long)(((u64)(u32)wave_bcast((int)(u32)((u64)d >> 32), src) << 32) | (u32)wave_bcast((int)(u32)d, src));
            for (int i = lane; i < rn; i += 64) col_idx[rd + i] = tmp[rs + i];
        }
    }
}

void launch_small(const int *Arow, const int *Acol, const int *Brow, const int *Bcol, int row_begin, int nrows,
                  long long *F, long long *Fprefix, int *list, int *cnt, int *tmp, long long *row_ptr, int *col_idx,
                  SmallTiles *tl, SmallScalars *sc, hipStream_t s)
{
    const int g32 = (nrows + 31) / 32, g256 = (nrows + 255) / 256;
    hipLaunchKernelGGL(k_small_sizes, dim3(g32), dim3(256), 0, s, Arow, Acol, Brow, row_begin, nrows, F, tl);
    hipLaunchKernelGGL(k_small_plan, dim3(g256), dim3(256), 0, s, F, nrows, Arow, row_begin, Fprefix, list, cnt, tl, sc);
    // persistent grids over the list (at most kSmallMaxRows rows = 2048 batches of 64)
    hipLaunchKernelGGL(k_small_rows, dim3(256), dim3(64 * kSmallWaves), 0, s, Arow, Acol, Brow, Bcol, row_begin, list, Fprefix,
                       sc, tmp, cnt, tl);
    hipLaunchKernelGGL(k_small_scan, dim3(g256), dim3(256), 0, s, cnt, nrows, row_ptr, tl, sc);
    hipLaunchKernelGGL(k_small_copy, dim3(256), dim3(256), 0, s, cnt, Fprefix, list, tmp, row_ptr, col_idx, sc);
}

}  // namespace bsp
