// pack_tiles.hip -- cuts the rows of a product into TILES for the fused flow (tile_rows.inc).
//
// The reference appends row after row to one growing array (final/SpGEMM_mpi_omp.c:28-42): row i
// lands where row i-1 ended.  The fused flow keeps exactly that order on the GPU: consecutive
// rows are packed into tiles of at most `cap` products and `maxr` rows, one workgroup accumulates
// a whole tile at once and a look-back chain over the tiles hands every tile the place where the
// previous one ended.  This file only decides the cuts:
//   * greedy packing -- a tile takes rows while its products stay <= cap -- so that a tile is
//     nearly full whatever the row lengths are (a fixed rule like "cut at multiples of cap" leaves
//     tiles half empty because a row may not be split);
//   * greedy is a sequential recurrence (next cut = f(this cut)); it is evaluated in parallel by
//     pointer doubling inside blocks of 2048 rows, each block starting a fresh tile (one short tile
//     per 2048 rows);
//   * a row with more than `cap` products is a tile of its own, flagged: the heavy-row kernels
//     (dense_rows.hip) have computed it into the workspace before the tile kernel runs, the tile
//     only carries its count through the chain.
// Two passes (count -> scan of the per-block counts -> emit) because a tile's index is its rank in
// row order.  HBM traffic: F is read twice (16 B per row), 16 B written per tile.
#include "kernels.hpp"
#include "wave.hpp"

namespace bsp {

constexpr int kPackRows = 2048;          // rows per packing block
constexpr int kPackThreads = 256;
constexpr int kPackItems = kPackRows / kPackThreads;

// Shared by both passes: weights w_i = min(F_i, cap + 1) of the block's rows -> exclusive prefix P[]
// in LDS (P[nloc] = total); returns this thread's 8 weights.
__device__ __forceinline__ void pack_prefix(const long long *__restrict__ F, int n, int cap, int *P, int *wsum,
                                            long long *bound_sum, int cols)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int base = blockIdx.x * kPackRows + tid * kPackItems;
    int wt[kPackItems];
    int run = 0;
    long long bsum = 0;
#pragma unroll
    for (int k = 0; k < kPackItems; k++) {
        long long f = (base + k < n) ? F[base + k] : 0;
        bsum += (cols > 0 && f > cols) ? (long long)cols : f;
        wt[k] = f > cap ? cap + 1 : (int)f;
        run += wt[k];
    }
    const int inc = wave_incl_scan(run);
    if (lane == 63) wsum[w] = inc;
    if (bound_sum) {
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) bsum += __shfl_xor(bsum, d, 64);
        if (lane == 0) bound_sum[w] = bsum;
    }
    __syncthreads();
    int off = inc - run;
    for (int k = 0; k < w; k++) off += wsum[k];
#pragma unroll
    for (int k = 0; k < kPackItems; k++) {
        P[tid * kPackItems + k] = off;
        off += wt[k];
    }
    if (tid == kPackThreads - 1) P[kPackRows] = off;
    __syncthreads();
}

// end of the tile that starts at local row i: the largest e in (i, min(i + maxr, nloc)] with
// P[e] - P[i] <= cap, at least i + 1 (a row above cap is a tile of its own)
__device__ __forceinline__ int pack_next(const int *P, int i, int nloc, int cap, int maxr)
{
    int lo = i + 1;                                   // always feasible as a cut (single row)
    int hi = i + maxr < nloc ? i + maxr : nloc;       // candidates (lo, hi]
    const int lim = P[i] + cap;
    if (P[lo] > lim) return lo;                       // the row itself exceeds cap
    while (lo < hi) {                                 // invariant: P[lo] <= lim
        const int mid = (lo + hi + 1) >> 1;
        if (P[mid] <= lim) lo = mid; else hi = mid - 1;
    }
    return lo;
}

// pass 1: marks8[row / 8] = bitmap of the rows that start a tile; tile_count[block]; bound[block] =
// sum of min(F_i, cols) over the block's rows (what C.col_idx is sized by)
__global__ __launch_bounds__(kPackThreads) void k_pack_count(const long long *__restrict__ F, int n, int cap, int maxr,
                                                             int cols, unsigned char *__restrict__ marks8,
                                                             int *__restrict__ tile_count,
                                                             long long *__restrict__ bound)
{
    __shared__ int P[kPackRows + 1];
    __shared__ unsigned short jmp[2][kPackRows];
    __shared__ unsigned char mark[kPackRows];
    __shared__ int wsum[4];
    __shared__ long long bsum[4];
    const int tid = threadIdx.x;
    const int row0 = blockIdx.x * kPackRows;
    const int nloc = (n - row0 < kPackRows) ? n - row0 : kPackRows;
    pack_prefix(F, n, cap, P, wsum, bsum, cols);
#pragma unroll
    for (int k = 0; k < kPackItems; k++) {
        const int i = tid * kPackItems + k;
        jmp[0][i] = (unsigned short)(i < nloc ? pack_next(P, i, nloc, cap, maxr) : kPackRows);
        mark[i] = (i == 0) ? 1 : 0;
    }
    __syncthreads();
    // orbit of row 0 under next(): after round r every orbit member within 2^(r+1) steps is marked
    int cur = 0;
    for (int r = 0; r < 11; r++) {
        unsigned char m[kPackItems];
        unsigned short j1[kPackItems];
#pragma unroll
        for (int k = 0; k < kPackItems; k++) {
            const int i = tid * kPackItems + k;
            m[k] = mark[i];
            j1[k] = jmp[cur][i];
        }
        unsigned short j2[kPackItems];
#pragma unroll
        for (int k = 0; k < kPackItems; k++) j2[k] = j1[k] < kPackRows ? jmp[cur][j1[k]] : (unsigned short)kPackRows;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < kPackItems; k++) {
            const int i = tid * kPackItems + k;
            if (m[k] && j1[k] < nloc) mark[j1[k]] = 1;
            jmp[cur ^ 1][i] = j2[k];
        }
        cur ^= 1;
        __syncthreads();
    }
    unsigned bits = 0u;
#pragma unroll
    for (int k = 0; k < kPackItems; k++) {
        const int i = tid * kPackItems + k;
        if (i < nloc && mark[i]) bits |= 1u << k;
    }
    static_assert(kPackItems == 8, "one byte of marks per thread");
    marks8[(size_t)blockIdx.x * kPackThreads + tid] = (unsigned char)bits;
    int c = __popc(bits);
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) c += __shfl_xor(c, d, 64);
    __syncthreads();
    if ((tid & 63) == 0) wsum[tid >> 6] = c;
    __syncthreads();
    if (tid == 0) {
        tile_count[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        bound[blockIdx.x] = bsum[0] + bsum[1] + bsum[2] + bsum[3];
    }
}

// one workgroup: tile_count[] -> exclusive prefix in place, totals[0] = number of tiles,
// totals[1] = sum of bound[]
__global__ __launch_bounds__(1024) void k_pack_scan(int *__restrict__ tile_count, const long long *__restrict__ bound, int m,
                                                    long long *__restrict__ totals)
{
    __shared__ long long wsum[16], bs[16];
    __shared__ long long carry, bcarry;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    if (tid == 0) { carry = 0; bcarry = 0; }
    __syncthreads();
    for (int base = 0; base < m; base += 1024) {
        const int i = base + tid;
        const long long v = i < m ? (long long)tile_count[i] : 0;
        long long b = i < m ? bound[i] : 0;
        const long long inc = wave_incl_scan64(v);
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) b += __shfl_xor(b, d, 64);
        if (lane == 63) wsum[w] = inc;
        if (lane == 0) bs[w] = b;
        __syncthreads();
        long long woff = 0, total = 0, btotal = 0;
        for (int k = 0; k < 16; k++) {
            if (k < w) woff += wsum[k];
            total += wsum[k];
            btotal += bs[k];
        }
        const long long c = carry;
        if (i < m) tile_count[i] = (int)(c + woff + inc - v);
        __syncthreads();
        if (tid == 0) { carry = c + total; bcarry += btotal; }
        __syncthreads();
    }
    if (tid == 0) { totals[0] = carry; totals[1] = bcarry; }
}

// pass 2: the tile descriptors, in row order
__global__ __launch_bounds__(kPackThreads) void k_pack_emit(const long long *__restrict__ F, int n, int cap, int maxr,
                                                            const int *__restrict__ Arow,     // A.row_ptr + row_begin
                                                            const unsigned char *__restrict__ marks8,
                                                            const int *__restrict__ tile_base,
                                                            TileDesc *__restrict__ tiles)
{
    __shared__ int P[kPackRows + 1];
    __shared__ int wsum[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int row0 = blockIdx.x * kPackRows;
    const int nloc = (n - row0 < kPackRows) ? n - row0 : kPackRows;
    pack_prefix(F, n, cap, P, wsum, nullptr, 0);
    const unsigned bits = marks8[(size_t)blockIdx.x * kPackThreads + tid];
    const int mine = __popc(bits);
    const int inc = wave_incl_scan(mine);
    __syncthreads();                                  // wsum is reused
    if (lane == 63) wsum[w] = inc;
    __syncthreads();
    int pos = tile_base[blockIdx.x] + inc - mine;
    for (int k = 0; k < w; k++) pos += wsum[k];
#pragma unroll
    for (int k = 0; k < kPackItems; k++) {
        if (!((bits >> k) & 1u)) continue;
        const int i = tid * kPackItems + k;
        const int e = pack_next(P, i, nloc, cap, maxr);
        const int f = P[e] - P[i];
        TileDesc d;
        d.row0 = row0 + i;
        const bool heavy = f > cap;                   // only possible for a single row
        d.nrf = (unsigned)(e - i) | ((heavy ? 0u : (unsigned)f) << 8) | (heavy ? 0x80000000u : 0u);
        d.a0 = Arow[row0 + i];
        d.nsrc = Arow[row0 + e] - d.a0;
        tiles[pos++] = d;
    }
}

void launch_pack_tiles_count(const long long *F, int n, int cap, int maxr, int cols, unsigned char *marks8,
                             int *tile_count, long long *bound, long long *totals, hipStream_t s)
{
    if (n <= 0) {
        hipMemsetAsync(totals, 0, 2 * sizeof(long long), s);
        return;
    }
    const int blocks = (n + kPackRows - 1) / kPackRows;
    hipLaunchKernelGGL(k_pack_count, dim3(blocks), dim3(kPackThreads), 0, s, F, n, cap, maxr, cols, marks8, tile_count, bound);
    hipLaunchKernelGGL(k_pack_scan, dim3(1), dim3(1024), 0, s, tile_count, bound, blocks, totals);
}

void launch_pack_tiles_emit(const long long *F, int n, int cap, int maxr, const int *Arow, const unsigned char *marks8,
                            const int *tile_base, TileDesc *tiles, hipStream_t s)
{
    if (n <= 0) return;
    const int blocks = (n + kPackRows - 1) / kPackRows;
    hipLaunchKernelGGL(k_pack_emit, dim3(blocks), dim3(kPackThreads), 0, s, F, n, cap, maxr, Arow, marks8, tile_base, tiles);
}

}  // namespace bsp
