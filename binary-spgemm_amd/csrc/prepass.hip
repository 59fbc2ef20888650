// prepass.hip -- symbolic side of bspgemm_multiply: per-row work, scans, capacity binning.
//
// The reference has no counterpart: SpGEMM_bigslice (final/SpGEMM_mpi_omp.c:15-58) discovers
// a row's size while it fills it and grows Ccol with realloc (:28-31).  On the GPU every row is
// sized up front by its product count F_i = sum_{j in A_i} |B_j| (an upper bound of |C_i|), which
// (a) fixes where each row's result lands without any inter-wave dependency and (b) selects
// the accumulator capacity class.
//
// The prepass also flattens the pointer chasing the accumulate kernels would otherwise do per
// row (rows[k] -> A.row_ptr -> A.col_idx -> B.row_ptr -> B.col_idx, five dependent loads):
//   * k_row_work leaves, for every A-nonzero jj, the pair ab[jj] = (B.row_ptr[j], |B_j|), so
//     the hot kernel reads the B-row extents coalesced instead of gathering them;
//   * k_scan_apply leaves one record per non-empty row, grouped by capacity class (bins are
//     exact segments of one array, rows in ascending order up to 2048-row tiles):
//     {row, A.row_ptr[row], |A_row|, F_row} + the row's output offset.
// All kernels here are HBM-bound streaming/gather passes: row_work reads 4*nnzA (A.col_idx) +
// 8*nnzA (B.row_ptr pairs, random 8-byte gathers) and writes 8*nnzA; the scans touch ~40 B/row.
#include "kernels.hpp"
#include "wave.hpp"

namespace bsp {

struct __attribute__((packed, aligned(4))) Int2U { int x, y; };   // 8 B, only dword aligned

// A wave holds 8 rows, 8 lanes each.  Rows of more than kLongRow nonzeros are first swept by ALL 64
// lanes, one row at a time (a hub row of 36000 nonzeros was 1136 dependent trips of its 8 lanes: the
// whole prepass of a power-law input waited for it); the others by their own 8 lanes.  `sweep(first, end,
// stride)` handles nonzeros first, first+stride, ... and returns the lane's partial sum; the result is the
// row's sum in the row's lane 0 (sub == 0).
constexpr int kLongRow = 256;
constexpr int kRowWorkUnroll = 4;            // nonzeros per lane and trip (independent gathers in flight)
template <typename Sweep>
__device__ __forceinline__ long long rows_of_a_wave(int a0, int a1, int sub, Sweep &&sweep)
{
    const int lane = threadIdx.x & 63;
    long long long_sum = 0;
#pragma unroll
    for (int q = 0; q < 8; q++) {
        const int qa0 = __builtin_amdgcn_readlane(a0, q * 8), qa1 = __builtin_amdgcn_readlane(a1, q * 8);
        if (qa1 - qa0 > kLongRow) {                                // wave-uniform
            long long t = sweep((long long)qa0 + lane, (long long)qa1, 64);
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) t += __shfl_xor(t, d, 64);
            if (lane == q * 8) long_sum = t;
        }
    }
    const bool is_long = a1 - a0 > kLongRow;
    long long sum = is_long ? 0ll : sweep((long long)a0 + sub, (long long)a1, 8);
    sum += __shfl_xor(sum, 1, 64);
    sum += __shfl_xor(sum, 2, 64);
    sum += __shfl_xor(sum, 4, 64);
    return is_long ? long_sum : sum;
}

// ---------------------------------------------------------------------------------------
// 8 lanes per A row: lanes stride the row's col_idx (coalesced 32-B pieces), gather the
// B.row_ptr pair of every nonzero, reduce over the 8 lanes.
// PAD: B's rows live in the padded copy of B.col_idx (every row on a 64-byte boundary, bspgemm_matrix::d_col_pad); what is
// gathered per A-nonzero is then the row's entry {start in the padded copy, length} of the extents table Bext -- one aligned
// 8-byte read, like the B.row_ptr pair it replaces.
template <bool PAD>
__global__ __launch_bounds__(256) void k_row_work(const int *__restrict__ Arow,
                                                  const int *__restrict__ Acol,
                                                  const int *__restrict__ Brow,
                                                  const int2 *__restrict__ Bext,
                                                  int row_begin, int nrows,
                                                  long long *__restrict__ F,
                                                  int2 *__restrict__ ab)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(gid >> 3);
    const int sub = (int)(gid & 7);
    int a0 = 0, a1 = 0;
    if (r < nrows) { a0 = Arow[row_begin + r]; a1 = Arow[row_begin + r + 1]; }
    // U nonzeros per lane and trip: their A.col_idx loads, then their B.row_ptr gathers, are
    // issued together (independent misses in flight instead of one dependent chain per nonzero)
    constexpr int U = kRowWorkUnroll;
    auto sweep = [&](long long first, long long end, int stride) {
        long long sum = 0;
        for (long long jj = first; jj < end; jj += (long long)stride * U) {   // 64-bit: jj + stride*u may pass INT_MAX
            int j[U];
#pragma unroll
            for (int u = 0; u < U; u++) j[u] = (jj + stride * u < end) ? Acol[jj + stride * u] : -1;
            Int2U pr[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                pr[u].x = pr[u].y = 0;
                if (j[u] >= 0) {
                    if (PAD) { const int2 e = Bext[j[u]]; pr[u].x = e.x; pr[u].y = e.x + e.y; }
                    else pr[u] = *reinterpret_cast<const Int2U *>(Brow + j[u]);           // one 8-B gather (dword aligned)
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (j[u] >= 0) {
                    const int len = pr[u].y - pr[u].x;
                    ab[jj + stride * u] = make_int2(pr[u].x, len);
                    sum += (long long)len;
                }
        }
        return sum;
    };
    const long long sum = rows_of_a_wave(a0, a1, sub, sweep);
    if (r < nrows && sub == 0) F[r] = sum;
}

// ---------------------------------------------------------------------------------------
// The same pass through B's BLOCKED extents table: per 8 rows of B one 12-byte entry
// {row_ptr of the block's first row, the 8 row lengths clamped to 255}.  The table is 1.5 B per
// row (6.3 MB at 4.2 M rows against 16.8 MB of B.row_ptr), so most of it stays in an XCD's 4 MB L2
// on a skewed input and the 64-byte sectors that do come from the fabric are shared by 5 blocks
// = 40 rows instead of 16.  start = base + sum of the lengths below the row in its block (two
// v_sad_u8), length = the row's byte; a block with a clamped byte at or below the row is looked up
// in B.row_ptr itself (rows of 255+ nonzeros: rare).
struct __attribute__((packed, aligned(4))) Blk8 { int base; unsigned lo, hi; };   // 12 B, only dword aligned

__global__ __launch_bounds__(256) void k_blk8(const int *__restrict__ row_ptr, const int *__restrict__ start_ptr, int n,
                                              int *__restrict__ blk, unsigned long long *__restrict__ clamped_nnz)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b * 8 >= n) return;
    const int r0 = b * 8;
    unsigned lo = 0u, hi = 0u;
    long long clamped = 0;
    int prev = row_ptr[r0];
    const int base = start_ptr[r0];                                // where the block's first row starts (B.col_idx or its padded copy)
#pragma unroll
    for (int k = 0; k < 8; k++) {
        int d = 0;
        if (r0 + k < n) {
            const int nx = row_ptr[r0 + k + 1];
            d = nx - prev;
            prev = nx;
        }
        const unsigned byte = (unsigned)(d < 255 ? d : 255);
        if (d >= 255) clamped += d;
        if (k < 4) lo |= byte << (8 * k); else hi |= byte << (8 * (k - 4));
    }
    // nonzeros that live in clamped rows: how skewed the operand is (the caller's switch, see api.hip)
    if (clamped > 0) atomicAdd(clamped_nnz, (unsigned long long)clamped);
    blk[3 * b] = base;
    blk[3 * b + 1] = (int)lo;
    blk[3 * b + 2] = (int)hi;
}
void launch_blk8(const int *row_ptr, const int *start_ptr, int n, int *blk, unsigned long long *clamped_nnz, hipStream_t s)
{
    if (n <= 0) return;
    const int nb = (n + 7) / 8;
    hipLaunchKernelGGL(k_blk8, dim3((nb + 255) / 256), dim3(256), 0, s, row_ptr, start_ptr ? start_ptr : row_ptr, n, blk, clamped_nnz);
}

// PAD: the table's `base` is the block's first row in the PADDED copy of B.col_idx and a row starts behind the padded
// lengths below it -- ceil(len / 16) * 16, summed over the bytes in SWAR form; rows behind a clamped byte are looked up in
// the padded row_ptr (Bpad) and B.row_ptr.
template <bool PAD>
__global__ __launch_bounds__(256) void k_row_work_blk(const int *__restrict__ Arow,
                                                      const int *__restrict__ Acol,
                                                      const int *__restrict__ Brow,
                                                      const int *__restrict__ Bblk,
                                                      const int *__restrict__ Bpad,
                                                      int row_begin, int nrows,
                                                      long long *__restrict__ F,
                                                      int2 *__restrict__ ab)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(gid >> 3);
    const int sub = (int)(gid & 7);
    int a0 = 0, a1 = 0;
    if (r < nrows) { a0 = Arow[row_begin + r]; a1 = Arow[row_begin + r + 1]; }
    constexpr int U = kRowWorkUnroll;
    auto sweep = [&](long long first, long long end, int stride) {
        long long sum = 0;
        for (long long jj = first; jj < end; jj += (long long)stride * U) {
            int j[U];
#pragma unroll
            for (int u = 0; u < U; u++) j[u] = (jj + stride * u < end) ? Acol[jj + stride * u] : -1;
            Blk8 w[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                w[u].base = 0; w[u].lo = w[u].hi = 0u;
                if (j[u] >= 0) w[u] = *reinterpret_cast<const Blk8 *>(Bblk + 3 * (j[u] >> 3));
            }
            int start[U], len[U];
            bool sat[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                const int k = j[u] & 7;
                const u64 d = ((u64)w[u].hi << 32) | (u64)w[u].lo;
                const u64 below = d & ((1ull << (8 * k)) - 1ull);
                const u64 upto = (k == 7) ? d : (d & ((1ull << (8 * k + 8)) - 1ull));
                const u64 v = ~upto;                                // a 255 byte at or below the row -> a zero byte here
                sat[u] = j[u] >= 0 && ((v - 0x0101010101010101ull) & ~v & 0x8080808080808080ull) != 0ull;
                if (PAD) {
                    // sixteenths of each length below the row, rounded up: (len >> 4) + ((len & 15) != 0), byte by byte
                    const u64 m0f = 0x0f0f0f0f0f0f0f0full, m01 = 0x0101010101010101ull;
                    const u64 units = ((below >> 4) & m0f) + ((((below & m0f) + m0f) >> 4) & m01);
                    start[u] = w[u].base + 16 * ((int)__builtin_amdgcn_sad_u8((unsigned)units, 0u, 0u)
                                                 + (int)__builtin_amdgcn_sad_u8((unsigned)(units >> 32), 0u, 0u));
                } else {
                    start[u] = w[u].base + (int)__builtin_amdgcn_sad_u8((unsigned)below, 0u, 0u)
                               + (int)__builtin_amdgcn_sad_u8((unsigned)(below >> 32), 0u, 0u);
                }
                len[u] = (int)((d >> (8 * k)) & 255ull);
            }
            // clamped lengths (B rows of 255+ nonzeros -- the hubs of a skewed graph, so these reads hit
            // L2): the exact pairs, again issued together
            Int2U pr[U];
            int ps[U];
#pragma unroll
            for (int u = 0; u < U; u++) {
                pr[u].x = pr[u].y = 0;
                ps[u] = 0;
                if (sat[u]) {
                    pr[u] = *reinterpret_cast<const Int2U *>(Brow + j[u]);
                    if (PAD) ps[u] = Bpad[j[u]];
                }
            }
#pragma unroll
            for (int u = 0; u < U; u++)
                if (j[u] >= 0) {
                    if (sat[u]) { start[u] = PAD ? ps[u] : pr[u].x; len[u] = pr[u].y - pr[u].x; }
                    ab[jj + stride * u] = make_int2(start[u], len[u]);
                    sum += (long long)len[u];
                }
        }
        return sum;
    };
    const long long sum = rows_of_a_wave(a0, a1, sub, sweep);
    if (r < nrows && sub == 0) F[r] = sum;
}

void launch_row_work(const int *Arow, const int *Acol, const int *Brow, const int *Bblk8, const int *Bpad, const int2 *Bext,
                     int row_begin, int row_end, long long *F, int2 *ab, hipStream_t s)
{
    const int nrows = row_end - row_begin;
    if (nrows <= 0) return;
    const long long threads = (long long)nrows * 8;
    const int grid = (int)((threads + 255) / 256);
    if (Bblk8 && Bpad)
        hipLaunchKernelGGL(k_row_work_blk<true>, dim3(grid), dim3(256), 0, s, Arow, Acol, Brow, Bblk8, Bpad, row_begin, nrows, F, ab);
    else if (Bblk8)
        hipLaunchKernelGGL(k_row_work_blk<false>, dim3(grid), dim3(256), 0, s, Arow, Acol, Brow, Bblk8, nullptr, row_begin, nrows, F, ab);
    else if (Bext)
        hipLaunchKernelGGL(k_row_work<true>, dim3(grid), dim3(256), 0, s, Arow, Acol, Brow, Bext, row_begin, nrows, F, ab);
    else
        hipLaunchKernelGGL(k_row_work<false>, dim3(grid), dim3(256), 0, s, Arow, Acol, Brow, nullptr, row_begin, nrows, F, ab);
}

// ---------------------------------------------------------------------------------------
// The padded copy of an operand's col_idx: row j moves to pad_ptr[j], a multiple of 16 entries = a 64-byte boundary.
// A B row read by the accumulate kernels then touches ceil(len / 16) 64-byte sectors instead of one more (the gather of the
// bench matrix moves 7.3 GB instead of 9.4 GB: tools/gather_traffic_model.py).  plen[j] = padded length, scanned by
// launch_scan_counts into pad_ptr64; narrowed to int32 once the total is known to fit.
__global__ __launch_bounds__(256) void k_pad_lengths(const int *__restrict__ row_ptr, int n, int *__restrict__ plen)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) plen[i] = ((row_ptr[i + 1] - row_ptr[i]) + 15) & ~15;
}
void launch_pad_lengths(const int *row_ptr, int n, int *plen, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_pad_lengths, dim3((n + 255) / 256), dim3(256), 0, s, row_ptr, n, plen);
}
// 16 lanes per row: entries copied, the pad slots of the row's last sector filled with its last column (never read as
// products -- a lane past the end of a row is masked by its product index -- but defined); ext[j] = {pad_ptr[j], len}
__global__ __launch_bounds__(256) void k_pad_copy(const int *__restrict__ row_ptr, const int *__restrict__ col_idx,
                                                  const int *__restrict__ pad_ptr, int n, int *__restrict__ col_pad,
                                                  int2 *__restrict__ ext)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(gid >> 4), sub = (int)(gid & 15);
    if (r >= n) return;
    const int s0 = row_ptr[r], len = row_ptr[r + 1] - s0, d0 = pad_ptr[r];
    const int plen = (len + 15) & ~15;
    for (int t = sub; t < plen; t += 16) col_pad[d0 + t] = col_idx[s0 + (t < len ? t : len - 1)];
    if (sub == 0 && ext) ext[r] = make_int2(d0, len);
}
void launch_pad_copy(const int *row_ptr, const int *col_idx, const int *pad_ptr, int n, int *col_pad, int2 *ext, hipStream_t s)
{
    if (n <= 0) return;
    const long long threads = (long long)n * 16;
    hipLaunchKernelGGL(k_pad_copy, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, row_ptr, col_idx, pad_ptr, n, col_pad, ext);
}

// ---------------------------------------------------------------------------------------
// Products per row WITHOUT the extents: where only F_i is needed (shard cuts, the fused flow's tile
// packing, the closure's sizing).  What is gathered per A-nonzero is ONE BYTE from a table that fits an XCD's
// L2 for matrices up to ~4 M rows -- B's row lengths clamped to 255, built once per operand --
// instead of an 8-byte B.row_ptr pair out of a table four to eight times the L2 (the pair gather
// is what bounds k_row_work: one 64-byte sector fetched per nonzero).  A clamped entry (a B row
// of 255 or more) is looked up exactly.
__global__ __launch_bounds__(256) void k_deg8(const int *__restrict__ row_ptr, int n, unsigned char *__restrict__ deg8)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const int d = row_ptr[i + 1] - row_ptr[i];
        deg8[i] = (unsigned char)(d < 255 ? d : 255);
    }
}
void launch_deg8(const int *row_ptr, int n, unsigned char *deg8, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_deg8, dim3((n + 255) / 256), dim3(256), 0, s, row_ptr, n, deg8);
}

__global__ __launch_bounds__(256) void k_row_products(const int *__restrict__ Arow, const int *__restrict__ Acol,
                                                      const int *__restrict__ Brow,
                                                      const unsigned char *__restrict__ Bdeg8,
                                                      int row_begin, int nrows, long long *__restrict__ F)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(gid >> 3);
    const int sub = (int)(gid & 7);
    int a0 = 0, a1 = 0;
    if (r < nrows) { a0 = Arow[row_begin + r]; a1 = Arow[row_begin + r + 1]; }
    constexpr int U = 4;                                           // independent gathers in flight per lane
    auto sweep = [&](long long first, long long end, int stride) {
        long long sum = 0;
        for (long long jj = first; jj < end; jj += (long long)stride * U) {
            int j[U];
#pragma unroll
            for (int u = 0; u < U; u++) j[u] = (jj + stride * u < end) ? Acol[jj + stride * u] : -1;
            int d[U];
#pragma unroll
            for (int u = 0; u < U; u++) d[u] = j[u] >= 0 ? (int)Bdeg8[j[u]] : 0;
#pragma unroll
            for (int u = 0; u < U; u++) {
                if (d[u] == 255) d[u] = Brow[j[u] + 1] - Brow[j[u]];   // clamped: the exact length (rare)
                sum += (long long)d[u];
            }
        }
        return sum;
    };
    const long long sum = rows_of_a_wave(a0, a1, sub, sweep);
    if (r < nrows && sub == 0) F[r] = sum;
}

// Debug check behind BSPGEMM_OPT_CHECK: an operand's derived tables against its row_ptr.  A caller that rewrote a
// wrapped operand in place without bspgemm_matrix_invalidate would otherwise size rows from old lengths.
__global__ __launch_bounds__(256) void k_check_tables(const int *__restrict__ row_ptr, int n,
                                                      const unsigned char *__restrict__ deg8, const int *__restrict__ blk,
                                                      const int *__restrict__ pad_ptr, unsigned *__restrict__ err)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const int d = row_ptr[i + 1] - row_ptr[i];
    const unsigned want = (unsigned)(d < 255 ? d : 255);
    bool bad = d < 0;
    if (deg8) bad = bad || deg8[i] != want;
    if (blk) {
        const int b = i >> 3, k = i & 7;
        const unsigned w = (unsigned)blk[3 * b + 1 + (k >> 2)];
        bad = bad || ((w >> (8 * (k & 3))) & 255u) != want || (k == 0 && blk[3 * b] != (pad_ptr ? pad_ptr[i] : row_ptr[i]));
    }
    if (pad_ptr) bad = bad || pad_ptr[i + 1] - pad_ptr[i] != ((d + 15) & ~15);
    if (bad) atomicOr(err, kErrStaleTable);
}
void launch_check_tables(const int *row_ptr, int rows, const unsigned char *deg8, const int *blk8, const int *pad_ptr, unsigned *err,
                         hipStream_t s)
{
    if (rows <= 0 || (!deg8 && !blk8 && !pad_ptr)) return;
    hipLaunchKernelGGL(k_check_tables, dim3((rows + 255) / 256), dim3(256), 0, s, row_ptr, rows, deg8, blk8, pad_ptr, err);
}

void launch_row_products(const int *Arow, const int *Acol, const int *Brow, const unsigned char *Bdeg8,
                         int row_begin, int row_end, long long *F, hipStream_t s)
{
    const int nrows = row_end - row_begin;
    if (nrows <= 0) return;
    const long long threads = (long long)nrows * 8;
    hipLaunchKernelGGL(k_row_products, dim3((int)((threads + 255) / 256)), dim3(256), 0, s, Arow, Acol, Brow, Bdeg8,
                       row_begin, nrows, F);
}

// ---------------------------------------------------------------------------------------
// Exclusive scan in three kernels: tile sums -> scan of the tile sums -> apply.
constexpr int kScanThreads = 256;
constexpr int kScanItems = 8;
constexpr int kScanTile = kScanThreads * kScanItems;   // 2048 values per workgroup

// capacity class of a row with F products (see kernels.hpp)
__device__ __forceinline__ int bin_of(long long F, int mid_cap, int rank_cap)
{
    if (F <= 0) return 0;
    if (F > kMaxWaveCap) return F <= rank_cap ? kRankBin : (F > mid_cap ? kDenseBin : kMidBin);
    const int f = (int)F;
    int b = 1;
#pragma unroll
    for (int k = 1; k < kWaveBins; k++) b += (f > 64 * kWaveChunks[k]) ? 1 : 0;
    return b;
}

// `heavy_cols` > 0 (BIN only): hpartials[tile] = sum over the tile's heavy rows of min(F_i, heavy_cols),
// the room a heavy row can need in the heavy-row workspace (|C_i| <= min(F_i, cols))
template <typename T, bool BIN>
__global__ __launch_bounds__(kScanThreads) void k_tile_sums(const T *__restrict__ in, int n,
                                                            long long *__restrict__ partials,
                                                            int *__restrict__ bin_tiles,
                                                            int heavy_cols, long long *__restrict__ hpartials,
                                                            int mid_cap, int rank_cap, int bound_cols,
                                                            const long long *__restrict__ true_in,
                                                            long long *__restrict__ ppartials)
{
    __shared__ long long lds[4], ldh[4], ldp[4];
    __shared__ int lcount[kNumBins];
    if (BIN && threadIdx.x < kNumBins) lcount[threadIdx.x] = 0;
    if (BIN) __syncthreads();
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    long long v = 0, hv = 0, pv = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++)
        if (base + k < n) {
            const long long x = (long long)in[base + k];
            v += (BIN && bound_cols > 0 && x > bound_cols) ? (long long)bound_cols : x;   // what is PLACED: |C_i| <= min(F_i, cols)
            if (BIN && ppartials) pv += true_in ? true_in[base + k] : x;                 // the products themselves
            if (BIN) {
                const int b = bin_of(x, mid_cap, rank_cap);
                atomicAdd(&lcount[b], 1);
                if (b > kWaveBins) hv += x < heavy_cols ? x : heavy_cols;
            }
        }
    v = wave_incl_scan64(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) lds[w] = v;
    if (BIN && heavy_cols > 0) {
        hv = wave_incl_scan64(hv);
        if (lane == 63) ldh[w] = hv;
    }
    if (BIN && ppartials) {                                    // uniform
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) pv += __shfl_xor(pv, d, 64);
        if (lane == 0) ldp[w] = pv;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        partials[blockIdx.x] = lds[0] + lds[1] + lds[2] + lds[3];
        if (BIN && heavy_cols > 0) hpartials[blockIdx.x] = ldh[0] + ldh[1] + ldh[2] + ldh[3];
        if (BIN && ppartials) ppartials[blockIdx.x] = ldp[0] + ldp[1] + ldp[2] + ldp[3];
    }
    if (BIN && threadIdx.x < kNumBins) bin_tiles[blockIdx.x * kNumBins + threadIdx.x] = lcount[threadIdx.x];
}

// one workgroup: in-place exclusive scan of `m` values with stride; vals[m*stride] = total
template <typename V>
__device__ __forceinline__ void scan_column(V *vals, int m, int stride, long long *wsum, long long *carry_s)
{
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) *carry_s = 0;
    __syncthreads();
    for (int base = 0; base < m; base += 1024) {
        const int i = base + threadIdx.x;
        const long long v = i < m ? (long long)vals[(size_t)i * stride] : 0;
        const long long inc = wave_incl_scan64(v);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        long long woff = 0, total = 0;
        for (int k = 0; k < 16; k++) {
            const long long t = wsum[k];
            if (k < w) woff += t;
            total += t;
        }
        const long long carry = *carry_s;
        if (i < m) vals[(size_t)i * stride] = (V)(carry + woff + inc - v);
        __syncthreads();
        if (threadIdx.x == 0) *carry_s = carry + total;
        __syncthreads();
    }
    if (threadIdx.x == 0) vals[(size_t)m * stride] = (V)*carry_s;
}

// block 0 scans the tile sums; blocks 1..kNumBins (BIN only) scan one capacity class's per-tile
// counts; block kNumBins+1 scans the heavy-row bounds (heavy-row workspace in use) or, `sum_only`,
// just adds up hpartials[] (there the per-tile product counts).  `scal` (may be NULL) collects what
// the host reads back after the prepass, so that it is one copy instead of five.
__global__ __launch_bounds__(1024) void k_scan_partials(long long *__restrict__ partials, int m,
                                                        int *__restrict__ bin_tiles,
                                                        int *__restrict__ bin_count,
                                                        long long *__restrict__ hpartials, int sum_only,
                                                        PrepScalars *__restrict__ scal)
{
    __shared__ long long wsum[16];
    __shared__ long long carry_s;
    if (blockIdx.x == 0) {
        scan_column<long long>(partials, m, 1, wsum, &carry_s);
    } else if (blockIdx.x == kNumBins + 1) {
        if (sum_only) {
            long long v = 0;
            for (int i = threadIdx.x; i < m; i += 1024) v += hpartials[i];
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) v += __shfl_xor(v, d, 64);
            if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = v;
            __syncthreads();
            if (threadIdx.x == 0) {
                long long t = 0;
                for (int k = 0; k < 16; k++) t += wsum[k];
                scal->products = t;
            }
        } else {
            scan_column<long long>(hpartials, m, 1, wsum, &carry_s);
            if (scal && threadIdx.x == 0) scal->heavy_total = hpartials[m];
        }
    } else {
        const int b = blockIdx.x - 1;
        scan_column<int>(bin_tiles + b, m, kNumBins, wsum, &carry_s);
        __syncthreads();
        if (threadIdx.x == 0) {
            const int c = bin_tiles[(size_t)m * kNumBins + b];
            bin_count[b] = c;
            if (scal) scal->bin_count[b] = c;
        }
    }
}

// `out` and `carry_in` may alias (a row range continuing the row_ptr of the rows before it hands
// out[0] in as the carry): neither is __restrict__.
template <typename T, bool BIN>
__global__ __launch_bounds__(kScanThreads) void k_scan_apply(const T *__restrict__ in, int n,
                                                             const long long *__restrict__ partials,
                                                             long long *out,
                                                             int row_begin,
                                                             const int *__restrict__ Arow,
                                                             const int *__restrict__ bin_tiles,
                                                             const int *__restrict__ bin_count,
                                                             RowRec *__restrict__ rec,
                                                             long long *__restrict__ recpre,
                                                             int *__restrict__ cnt,
                                                             const long long *carry_in,
                                                             int heavy_cols,
                                                             const long long *__restrict__ hpartials,
                                                             int mid_cap, int rank_cap, int bound_cols,
                                                             PrepScalars *__restrict__ scal,
                                                             int *__restrict__ chunk_row)
{
    __shared__ long long wsum[4], hsum[4];
    __shared__ int lcount[kNumBins];
    __shared__ int lbase[kNumBins];
    const int base = blockIdx.x * kScanTile + threadIdx.x * kScanItems;
    long long v[kScanItems];             // what rows are classified by (products)
    long long pl[kScanItems];            // what they are placed by: the same, or bounded by the column count
    long long tsum = 0, hmine = 0;
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        v[k] = (base + k < n) ? (long long)in[base + k] : 0;
        pl[k] = (BIN && bound_cols > 0 && v[k] > bound_cols) ? (long long)bound_cols : v[k];
        tsum += pl[k];
        if (BIN && v[k] > kMaxWaveCap) hmine += v[k] < heavy_cols ? v[k] : heavy_cols;
    }
    if (BIN && threadIdx.x < kNumBins) {
        // segment start of class b in the record array (class 0 = empty rows has no records)
        int start = 0;
        for (int b = 1; b < (int)threadIdx.x; b++) start += bin_count[b];
        lbase[threadIdx.x] = start + bin_tiles[blockIdx.x * kNumBins + threadIdx.x];
        lcount[threadIdx.x] = 0;
    }
    const long long inc = wave_incl_scan64(tsum);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 63) wsum[w] = inc;
    long long hoff = 0;
    if (BIN && heavy_cols > 0) {                               // uniform
        const long long hinc = wave_incl_scan64(hmine);
        if (lane == 63) hsum[w] = hinc;
        hoff = hinc - hmine;
    }
    __syncthreads();
    const long long base0 = carry_in ? *carry_in : 0;     // carry of the rows before this range
    long long off = base0 + partials[blockIdx.x] + inc - tsum;
    for (int k = 0; k < w; k++) off += wsum[k];
    if (BIN && heavy_cols > 0) {
        hoff += hpartials[blockIdx.x];
        for (int k = 0; k < w; k++) hoff += hsum[k];
    }
#pragma unroll
    for (int k = 0; k < kScanItems; k++) {
        if (base + k < n) {
            out[base + k] = off;
            if (!BIN && chunk_row && pl[k] > 0) {
                // the compaction chunks that START inside this row; the row of the last output closes the last chunk
                for (long long c = (off + kCompactGran - 1) / kCompactGran; c * kCompactGran < off + pl[k]; c++)
                    chunk_row[c] = base + k;
                const long long total = base0 + partials[gridDim.x];
                if (off + pl[k] == total) chunk_row[(total + kCompactGran - 1) / kCompactGran] = base + k;
            }
            if (BIN) {
                const int b = bin_of(v[k], mid_cap, rank_cap);
                if (b == 0) {
                    cnt[base + k] = 0;
                } else {
                    const int pos = lbase[b] + atomicAdd(&lcount[b], 1);
                    const int row = row_begin + base + k;
                    const int a0 = Arow[row];
                    RowRec q;
                    q.row = row;
                    q.a0 = a0;
                    q.alen = Arow[row + 1] - a0;
                    q.f = v[k] > 0x7fffffffll ? 0x7fffffff : (int)v[k];
                    rec[pos] = q;
                    // where the row is first written: its upper-bound offset, or (heavy-row workspace
                    // in use) the heavy rows' own offsets; one-wave rows then never use recpre
                    recpre[pos] = (heavy_cols > 0 && b > kWaveBins) ? hoff : off;
                    if (heavy_cols > 0 && b > kWaveBins) hoff += v[k] < heavy_cols ? v[k] : heavy_cols;
                }
            }
        }
        off += pl[k];
    }
    if (blockIdx.x == gridDim.x - 1 && threadIdx.x == 0) {
        const long long total = base0 + partials[gridDim.x];
        out[n] = total;
        if (BIN && scal) {
            scal->totalF = total;
            scal->a_lo = Arow[row_begin];
            scal->a_hi = Arow[row_begin + n];
        }
    }
}

void launch_scan_and_bin(const long long *F, int n, int row_begin, const int *Arow, long long *prefix,
                         long long *partials, int *bin_tiles, int *bin_count, RowRec *rec,
                         long long *recpre, int *cnt, int heavy_cols, long long *hpartials, int mid_cap, int rank_cap, int bound_cols,
                         hipStream_t s, PrepScalars *scal, const long long *true_F)
{
    if (n <= 0) {
        hipMemsetAsync(prefix, 0, sizeof(long long), s);
        hipMemsetAsync(bin_count, 0, kNumBins * sizeof(int), s);
        if (heavy_cols > 0) hipMemsetAsync(hpartials, 0, sizeof(long long), s);
        if (scal) hipMemsetAsync(scal, 0, sizeof(PrepScalars), s);
        return;
    }
    const int tiles = (n + kScanTile - 1) / kScanTile;
    // without a heavy-row workspace hpartials[] is free: it carries the per-tile product counts
    const bool count_products = scal && heavy_cols <= 0;
    hipLaunchKernelGGL((k_tile_sums<long long, true>), dim3(tiles), dim3(kScanThreads), 0, s, F, n, partials, bin_tiles,
                       heavy_cols, hpartials, mid_cap, rank_cap, bound_cols, true_F, count_products ? hpartials : nullptr);
    hipLaunchKernelGGL(k_scan_partials, dim3(heavy_cols > 0 || count_products ? 2 + kNumBins : 1 + kNumBins), dim3(1024), 0,
                       s, partials, tiles, bin_tiles, bin_count, hpartials, count_products ? 1 : 0, scal);
    hipLaunchKernelGGL((k_scan_apply<long long, true>), dim3(tiles), dim3(kScanThreads), 0, s, F, n, partials,
                       prefix, row_begin, Arow, bin_tiles, bin_count, rec, recpre, cnt, nullptr, heavy_cols, hpartials,
                       mid_cap, rank_cap, bound_cols, scal, nullptr);
}

// prefix[0..n] = base + exclusive scan of cnt[0..n); `base` (device, may be NULL = 0) may alias
// prefix[0]: a range of rows continues the row_ptr of the rows before it
void launch_scan_counts(const int *cnt, int n, long long *prefix, long long *partials,
                        const long long *base, hipStream_t s, int *chunk_row)
{
    if (n <= 0) {
        if (!base) hipMemsetAsync(prefix, 0, sizeof(long long), s);
        return;
    }
    const int tiles = (n + kScanTile - 1) / kScanTile;
    hipLaunchKernelGGL((k_tile_sums<int, false>), dim3(tiles), dim3(kScanThreads), 0, s, cnt, n, partials, nullptr, 0, nullptr, 0, 0, 0, nullptr, nullptr);
    hipLaunchKernelGGL(k_scan_partials, dim3(1), dim3(1024), 0, s, partials, tiles, nullptr, nullptr, nullptr, 0, nullptr);
    hipLaunchKernelGGL((k_scan_apply<int, false>), dim3(tiles), dim3(kScanThreads), 0, s, cnt, n, partials, prefix,
                       0, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, base, 0, nullptr, 0, 0, 0, nullptr,
                       base ? nullptr : chunk_row);
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

__global__ void k_narrow_row_ptr(const long long *__restrict__ src, int *__restrict__ dst, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = (int)src[i];
}
void launch_narrow_row_ptr(const long long *src, int *dst, int n, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_narrow_row_ptr, dim3((n + 255) / 256), dim3(256), 0, s, src, dst, n);
}

// 8 lanes per row copy the row shifted by its index; lane 0 appends the diagonal
__global__ __launch_bounds__(256) void k_add_diagonal(const int *__restrict__ Arow, const int *__restrict__ Acol,
                                                      int n, int *__restrict__ Trow, int *__restrict__ Tcol)
{
    const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int r = (int)(gid >> 3), sub = (int)(gid & 7);
    if (r > n) return;
    if (r == n) {
        if (sub == 0) Trow[n] = Arow[n] + n;
        return;
    }
    const int a0 = Arow[r], a1 = Arow[r + 1], t0 = a0 + r;
    for (int k = a0 + sub; k < a1; k += 8) Tcol[t0 + (k - a0)] = Acol[k];
    if (sub == 0) {
        Tcol[t0 + (a1 - a0)] = r;
        Trow[r] = t0;
    }
}
void launch_add_diagonal(const int *Arow, const int *Acol, int n, int *Trow, int *Tcol, hipStream_t s)
{
    const long long threads = ((long long)n + 1) * 8;
    hipLaunchKernelGGL(k_add_diagonal, dim3((int)((threads + 255) / 256)), dim3(256), 0, s, Arow, Acol, n, Trow, Tcol);
}

}  // namespace bsp
