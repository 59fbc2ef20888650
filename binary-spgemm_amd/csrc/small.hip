// small.hip -- the single-round-trip path for SMALL products (round 4).
//
// The general flows cost a product of any size ~25 kernel launches and two host round trips (the class counts must
// reach the host before the class kernels can be launched): 0.15 ms for the reference's own validity fixture
// (Matlab/validity_test.mtx: 12 502 products), which one CPU core multiplies in 0.79 ms -- launch latency,
// not work.  A product with few A-nonzeros takes this path instead: FIVE launches, no size ever goes to the host before
// the end, one read-back.
//     k_small_sizes   F_i = sum of |B_j| over A_i (8 lanes per row, B.row_ptr pairs read directly)
//     k_small_plan    one workgroup: exclusive scan of F, list of the non-empty rows, totals; decides whether the
//                     product FITS this path (F <= kSmallMaxProducts and every F_i <= kSmallMaxRow)
//     k_small_rows    one wave per non-empty row (persistent grid over the list): products gathered into LDS, bitonic
//                     sort, duplicates dropped, the row written at its upper-bound place -- the reference's own
//                     gather / sort / emit of one row (final/SpGEMM_mpi_omp.c:33-47) without the flag array
//     k_small_scan    one workgroup: scan of the row sizes = C.row_ptr, nnz(C)
//     k_small_copy    rows squeezed into C.col_idx (one wave per non-empty row)
// If the product does not fit, every kernel after the plan is a no-op, the host sees `bail` in the one read-back and
// runs the general flow (one wasted round trip).  Same CSR, bit for bit, as the other flows (tests/test_gpu_parity.py).
#include "kernels.hpp"
#include "wave.hpp"

namespace bsp {

// ---------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_small_sizes(const int *__restrict__ Arow, const int *__restrict__ Acol,
                                                     const int *__restrict__ Brow, int row_begin, int nrows,
                                                     long long *__restrict__ F)
{
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
}

// one workgroup of 1024 threads: thread t owns the rows [t * per, t * per + per)
__global__ __launch_bounds__(1024) void k_small_plan(const long long *__restrict__ F, int nrows, const int *__restrict__ Arow,
                                                     int row_begin, long long *__restrict__ Fprefix, int *__restrict__ list,
                                                     int *__restrict__ cnt, SmallScalars *__restrict__ sc)
{
    __shared__ long long s_sum[16];
    __shared__ int s_cnt[16], s_max[16];
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int per = (nrows + 1023) / 1024;
    const int r0 = t * per, r1 = min(r0 + per, nrows);
    long long mine = 0;
    int nz = 0;
    long long big = 0;
    for (int r = r0; r < r1; r++) {
        const long long f = F[r];
        mine += f;
        nz += f > 0;
        big = f > big ? f : big;
    }
    const long long inc = wave_incl_scan64(mine);
    const int cinc = wave_incl_scan(nz);
    int bigc = big > 0x7fffffffll ? 0x7fffffff : (int)big;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) bigc = max(bigc, __shfl_xor(bigc, d, 64));
    if (lane == 63) { s_sum[w] = inc; s_cnt[w] = cinc; }
    if (lane == 0) s_max[w] = bigc;
    __syncthreads();
    long long off = inc - mine, total = 0;
    int coff = cinc - nz, ctotal = 0, maxf = 0;
    for (int k = 0; k < 16; k++) {
        if (k < w) { off += s_sum[k]; coff += s_cnt[k]; }
        total += s_sum[k];
        ctotal += s_cnt[k];
        maxf = max(maxf, s_max[k]);
    }
    const bool fits = total <= kSmallMaxProducts && maxf <= kSmallMaxRow;
    for (int r = r0; r < r1; r++) {
        const long long f = F[r];
        Fprefix[r] = off;
        cnt[r] = 0;
        if (f > 0 && fits) list[coff++] = r;
        off += f;
    }
    if (t == 0) {
        Fprefix[nrows] = total;
        sc->totalF = total;
        sc->nonempty = ctotal;
        sc->bail = fits ? 0 : 1;
        sc->nnzC = 0;
        sc->a_lo = Arow[row_begin];
        sc->a_hi = Arow[row_begin + nrows];
    }
}

// ---------------------------------------------------------------------------------------
// One wave per listed row.  keys[] (LDS, kSmallMaxRow entries per wave) receives the row's products; a bitonic network
// over the next power of two sorts them (pad = 0xffffffff, above every column); neighbours that are equal are dropped.
constexpr int kSmallWaves = 4;
__global__ __launch_bounds__(64 * kSmallWaves) void k_small_rows(const int *__restrict__ Arow, const int *__restrict__ Acol,
                                                                 const int *__restrict__ Brow, const int *__restrict__ Bcol,
                                                                 int row_begin, const int *__restrict__ list,
                                                                 const long long *__restrict__ Fprefix,
                                                                 const SmallScalars *__restrict__ sc,
                                                                 int *__restrict__ tmp, int *__restrict__ cnt)
{
    __shared__ u32 s_keys[kSmallWaves][kSmallMaxRow];
    if (sc->bail) return;                                          // (uniform: written by the kernel before this one)
    const int nlist = sc->nonempty;
    const int lane = lane_id(), wv = threadIdx.x >> 6;
    u32 *keys = s_keys[wv];
    const int nwaves = gridDim.x * kSmallWaves;
    for (int k = blockIdx.x * kSmallWaves + wv; k < nlist; k += nwaves) {
        const int r = list[k];
        const int a0 = Arow[row_begin + r], a1 = Arow[row_begin + r + 1];
        const int F = (int)(Fprefix[r + 1] - Fprefix[r]);          // 1 .. kSmallMaxRow
        int N = 64;
        while (N < F) N <<= 1;
        for (int i = lane; i < N; i += 64) keys[i] = ~0u;
        wave_lds_fence();
        // gather: 64 sources at a time, lane s copies B row s to its place in product order
        int done = 0;
        for (int s0 = a0; s0 < a1; s0 += 64) {
            int b0 = 0, len = 0;
            if (s0 + lane < a1) {
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
        int *out = tmp + Fprefix[r];
        int written = 0;
        for (int i0 = 0; i0 < F; i0 += 64) {
            const int i = i0 + lane;
            const bool keep = i < F && (i == 0 || keys[i] != keys[i - 1]);
            const u64 bal = __ballot(keep);
            if (keep) out[written + __popcll(bal & mask_lt(lane))] = (int)keys[i];
            written += __popcll(bal);
        }
        if (lane == 0) cnt[r] = written;
        wave_lds_fence();
    }
}

// one workgroup: C.row_ptr = scan of cnt, nnz(C)
__global__ __launch_bounds__(1024) void k_small_scan(const int *__restrict__ cnt, int nrows, long long *__restrict__ row_ptr,
                                                     SmallScalars *__restrict__ sc)
{
    __shared__ long long s_sum[16];
    if (sc->bail) return;
    const int t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int per = (nrows + 1023) / 1024;
    const int r0 = t * per, r1 = min(r0 + per, nrows);
    long long mine = 0;
    for (int r = r0; r < r1; r++) mine += cnt[r];
    const long long inc = wave_incl_scan64(mine);
    if (lane == 63) s_sum[w] = inc;
    __syncthreads();
    long long off = inc - mine, total = 0;
    for (int k = 0; k < 16; k++) {
        if (k < w) off += s_sum[k];
        total += s_sum[k];
    }
    for (int r = r0; r < r1; r++) {
        row_ptr[r] = off;
        off += cnt[r];
    }
    if (t == 0) {
        row_ptr[nrows] = total;
        sc->nnzC = total;
    }
}

// rows from their upper-bound places to their final ones: one wave per listed row
__global__ __launch_bounds__(256) void k_small_copy(const int *__restrict__ cnt, const long long *__restrict__ Fprefix,
                                                    const int *__restrict__ list, const int *__restrict__ tmp,
                                                    const long long *__restrict__ row_ptr, int *__restrict__ col_idx,
                                                    const SmallScalars *__restrict__ sc)
{
    if (sc->bail) return;
    const int nlist = sc->nonempty;
    const int lane = threadIdx.x & 63;
    const int nwaves = gridDim.x * 4;
    for (int k = blockIdx.x * 4 + (threadIdx.x >> 6); k < nlist; k += nwaves) {
        const int r = list[k];
        const int n = cnt[r];
        const int *src = tmp + Fprefix[r];
        int *dst = col_idx + row_ptr[r];
        for (int i = lane; i < n; i += 64) dst[i] = src[i];
    }
}

void launch_small(const int *Arow, const int *Acol, const int *Brow, const int *Bcol, int row_begin, int nrows,
                  long long *F, long long *Fprefix, int *list, int *cnt, int *tmp, long long *row_ptr, int *col_idx,
                  SmallScalars *sc, hipStream_t s)
{
    const int threads = nrows * 8;
    hipLaunchKernelGGL(k_small_sizes, dim3((threads + 255) / 256), dim3(256), 0, s, Arow, Acol, Brow, row_begin, nrows, F);
    hipLaunchKernelGGL(k_small_plan, dim3(1), dim3(1024), 0, s, F, nrows, Arow, row_begin, Fprefix, list, cnt, sc);
    // persistent grid over the list: one workgroup per CU is plenty for at most kSmallMaxProducts products
    hipLaunchKernelGGL(k_small_rows, dim3(256), dim3(64 * kSmallWaves), 0, s, Arow, Acol, Brow, Bcol, row_begin, list, Fprefix,
                       sc, tmp, cnt);
    hipLaunchKernelGGL(k_small_scan, dim3(1), dim3(1024), 0, s, cnt, nrows, row_ptr, sc);
    hipLaunchKernelGGL(k_small_copy, dim3(256), dim3(256), 0, s, cnt, Fprefix, list, tmp, row_ptr, col_idx, sc);
}

}  // namespace bsp
