// wave_rows.hip -- the hot kernel: one wavefront per A-row, rank-bitmap accumulator in LDS.
//
// Replaces the body of SpGEMM_bigslice (final/SpGEMM_mpi_omp.c:24-52) for rows whose product
// count F_i fits one wave's capacity (<= 2048):
//     reference                                      here
//     xb[k] dense byte flags, one per column (:21)   hierarchy of 64-bit bitmaps in LDS
//     test-and-set + unsorted append (:38-42)        ds_or_b64 on the bitmaps (set union)
//     quickSort of the row (:47)                     none: slots are addressed by RANK, so the
//                                                    emit pass walks them in ascending order
//     sparse reset of xb (:48-50)                    slots/top words cleared as they are consumed
//
// Accumulator ("rank bitmap", an order-preserving perfect hash of the row's columns):
//   A column c is split into 6-bit digits.  The TOP bitmap is addressed directly by the high
//   digits (<= 128 words for any int32 column count).  Every set bit of a level owns one 64-bit
//   slot of the level below; the slot index is the bit's rank = prefix popcount, so slots are
//   dense (<= F_i of them, LDS use proportional to the row, independent of n) and in ascending
//   column order.  Level 0 slots are the 64-column masks of the result row.  Building it takes
//   LEVELS sweeps over the row's products, which stay in registers (col[], rank[]).
//   Per product: one B.col_idx load, LEVELS ds_or_b64, LEVELS-1 (ds_read_b64 + ds_read_u16).
//
// Gather: the F_i products of a row are the concatenation of the B rows selected by A's row.
//   Lanes first load the row's A.col_idx coalesced and the B.row_ptr pairs, a wave scan turns
//   the B row lengths into product offsets, a "starts" bitmap marks where each B row begins in
//   product order, and product p finds its source by popcount of the starts below p -- so all 64
//   lanes load B.col_idx every step whatever the B row lengths are.
//
// Output: row i's sorted columns go to tmp[Fprefix[i] ..), its count to cnt[i]; compact.hip
// squeezes the rows together once C.row_ptr is known.
//
// Roofline: HBM (gather of B.col_idx, 4 B per product, + 4 B per output written).  No MFMA.
#include "kernels.hpp"
#include "wave.hpp"

namespace bsp {

template <int LEVELS, int CHUNKS>
struct WaveLayout {
    static constexpr int CAP = 64 * CHUNKS;
    static constexpr int TOPW = (LEVELS == 4) ? 128 : 64;
    // byte offsets inside one wave's LDS slice (8-byte arrays first)
    static constexpr int oTop = 0;
    static constexpr int oStarts = oTop + 8 * TOPW;
    static constexpr int oSA = oStarts + 8 * CHUNKS;
    static constexpr int oSB = oSA + (LEVELS >= 2 ? 8 * CAP : 0);
    static constexpr int oL0w = oSB + (LEVELS >= 3 ? 8 * CAP : 0);
    static constexpr int oTopPre = oL0w + 4 * CAP;
    static constexpr int oPreA = oTopPre + 2 * TOPW;
    static constexpr int oPreB = oPreA + (LEVELS >= 4 ? 2 * CAP : 0);
    static constexpr int bytes = ((oPreB + (LEVELS >= 3 ? 2 * CAP : 0)) + 15) & ~15;
};

// scan the 64-bit words P[0..n) : pre[t] = number of set bits in P[0..t); returns the total
__device__ __forceinline__ int scan_words(const u64 *P, unsigned short *pre, int n, int lane)
{
    int running = 0;
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int t = t0 + lane;
        const u64 x = t < n ? P[t] : 0ull;
        const int c = __popcll(x);
        const int inc = wave_incl_scan(c);
        if (t < n) pre[t] = (unsigned short)(running + inc - c);
        running += wave_bcast(inc, 63);
    }
    return running;
}

__device__ __forceinline__ void clear_words(u64 *P, int n, int lane)
{
    for (int t = lane; t < n; t += 64) P[t] = 0ull;
}

template <int LEVELS, int CHUNKS>
__global__ __launch_bounds__(256) void k_wave_rows(const int *__restrict__ Arow,
                                                   const int *__restrict__ Acol,
                                                   const int *__restrict__ Brow,
                                                   const int *__restrict__ Bcol,
                                                   int topw,
                                                   const int *__restrict__ rows, int nrows,
                                                   int row_begin,
                                                   const long long *__restrict__ Fprefix,
                                                   int *__restrict__ tmp, int *__restrict__ cnt)
{
    using L = WaveLayout<LEVELS, CHUNKS>;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    const int lane = lane_id();
    const int wave_in_wg = threadIdx.x >> 6;
    unsigned char *my = lds_raw + wave_in_wg * L::bytes;
    u64 *top = reinterpret_cast<u64 *>(my + L::oTop);
    u64 *starts = reinterpret_cast<u64 *>(my + L::oStarts);
    u64 *SA = reinterpret_cast<u64 *>(my + L::oSA);
    u64 *SB = reinterpret_cast<u64 *>(my + L::oSB);
    u32 *L0w = reinterpret_cast<u32 *>(my + L::oL0w);
    int *delta = reinterpret_cast<int *>(my + L::oL0w);           // aliases L0w: dead before L0w lives
    unsigned short *topPre = reinterpret_cast<unsigned short *>(my + L::oTopPre);
    unsigned short *preA = reinterpret_cast<unsigned short *>(my + L::oPreA);
    unsigned short *preB = reinterpret_cast<unsigned short *>(my + L::oPreB);

    // zero the structures that must be all-zero at the start of a row (kept so by every row)
    for (int t = lane; t < L::TOPW; t += 64) top[t] = 0ull;
    if (lane < CHUNKS) starts[lane] = 0ull;
    if (LEVELS >= 2) clear_words(SA, L::CAP, lane);
    if (LEVELS >= 3) clear_words(SB, L::CAP, lane);
    wave_lds_fence();

    const long long wave_global = (long long)blockIdx.x * (blockDim.x >> 6) + wave_in_wg;
    const long long k0 = wave_global * kRowsPerWave;

    for (int kk = 0; kk < kRowsPerWave; kk++) {
        const long long k = k0 + kk;
        if (k >= nrows) break;                                     // wave-uniform
        const int i = rows[k];
        const int a0 = Arow[i], a1 = Arow[i + 1];

        // ---- gather plan: product offsets of the selected B rows ------------------------
        int F = 0, nsrc = 0;
        for (int ab = a0; ab < a1; ab += 64) {                     // usually one trip
            const int jj = ab + lane;
            int bs = 0, len = 0;
            if (jj < a1) {
                const int j = Acol[jj];
                bs = Brow[j];
                len = Brow[j + 1] - bs;
            }
            const int inc = wave_incl_scan(len);
            const int excl = F + inc - len;
            const u64 bal = __ballot(len > 0);
            if (len > 0) {
                const int sidx = nsrc + __popcll(bal & mask_lt(lane));
                delta[sidx] = bs - excl;                           // B address = delta + product index
                atomicOr(&starts[excl >> 6], 1ull << (excl & 63));
            }
            F += wave_bcast(inc, 63);
            nsrc += __popcll(bal);
        }
        wave_lds_fence();
        // starts words -> registers (lane c holds word c), then cleared for the next row
        u64 sw = 0ull;
        if (lane < CHUNKS) { sw = starts[lane]; starts[lane] = 0ull; }
        const int sinc = wave_incl_scan(__popcll(sw));
        const int sbefore = sinc - __popcll(sw);

        // ---- gather B.col_idx: all lanes busy, products kept in registers ---------------
        int col[CHUNKS];
        int rank[CHUNKS];
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            col[c] = -1;
            if (c * 64 < F) {                                      // wave-uniform
                const int p = c * 64 + lane;
                const u64 M = wave_bcast64(sw, c);
                const int before = wave_bcast(sbefore, c);
                if (p < F) {
                    const int s = before + __popcll(M & mask_le(lane)) - 1;
                    col[c] = Bcol[delta[s] + p];
                }
            }
        }
        wave_lds_fence();   // delta (aliases L0w) is dead from here on

        // ---- sweep 1: top bitmap, addressed directly by the high digits -----------------
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            if (col[c] >= 0) {
                const u32 cc = (u32)col[c];
                const u32 tw = cc >> (6 * LEVELS);
                atomicOr(&top[tw], 1ull << ((cc >> (6 * (LEVELS - 1))) & 63));
                rank[c] = (int)tw;
            }
        }
        wave_lds_fence();

        int n0;                      // number of level-0 slots (64-column masks) of this row
        u64 *S0;                     // where they live
        if (LEVELS == 1) {
            n0 = topw;
            S0 = top;
        } else {
            int nP = scan_words(top, topPre, topw, lane);          // slots of level LEVELS-2
            wave_lds_fence();
            const u64 *P = top;
            const unsigned short *Ppre = topPre;
            int nPwords = topw;
            u64 *Pmut = top;
#pragma unroll
            for (int lev = LEVELS - 2; lev >= 0; lev--) {
                // level `lev` slot buffers alternate: lev even -> SA, lev odd -> SB
                u64 *S = (lev & 1) ? SB : SA;
                unsigned short *Spre = (lev & 1) ? preB : preA;
#pragma unroll
                for (int c = 0; c < CHUNKS; c++) {
                    if (col[c] >= 0) {
                        const u32 cc = (u32)col[c];
                        const int r = rank[c];
                        const u64 x = P[r];
                        const int pre = Ppre[r];
                        const u32 b = (cc >> (6 * (lev + 1))) & 63;
                        const int r2 = pre + __popcll(x & ((1ull << b) - 1ull));
                        atomicOr(&S[r2], 1ull << ((cc >> (6 * lev)) & 63));
                        if (lev == 0) L0w[r2] = cc >> 6;
                        rank[c] = r2;
                    }
                }
                wave_lds_fence();
                clear_words(Pmut, nPwords, lane);                  // parent level is consumed
                if (lev > 0) {
                    const int nS = scan_words(S, Spre, nP, lane);
                    nPwords = nP;
                    nP = nS;
                    P = S;
                    Pmut = S;
                    Ppre = Spre;
                }
                wave_lds_fence();
            }
            n0 = nP;
            S0 = SA;                                               // level 0 is even
        }

        // ---- emit: walk the level-0 slots in rank order = ascending columns -------------
        int *out = tmp + Fprefix[i - row_begin];
        int running = 0;
        for (int t0 = 0; t0 < n0; t0 += 64) {
            const int t = t0 + lane;
            u64 m = 0ull;
            u32 w = 0;
            if (t < n0) {
                m = S0[t];
                w = (LEVELS == 1) ? (u32)t : L0w[t];
                S0[t] = 0ull;
            }
            const int c = __popcll(m);
            const int inc = wave_incl_scan(c);
            int pos = running + inc - c;
            const int base = (int)(w << 6);
            while (m) {
                out[pos++] = base | (int)__builtin_ctzll(m);
                m &= m - 1ull;
            }
            running += wave_bcast(inc, 63);
        }
        if (lane == 0) cnt[i - row_begin] = running;
        wave_lds_fence();
    }
}

template <int LEVELS, int CHUNKS>
static void launch_one(const int *Arow, const int *Acol, const int *Brow, const int *Bcol, int topw,
                       const int *rows, int nrows, int row_begin, const long long *Fprefix,
                       int *tmp, int *cnt, hipStream_t s)
{
    using L = WaveLayout<LEVELS, CHUNKS>;
    // waves per workgroup: 4, fewer when one wave's slice is large (LDS limit 64 KiB default)
    int waves = 4;
    while (waves > 1 && waves * L::bytes > 64 * 1024) waves >>= 1;
    const long long rows_per_wg = (long long)waves * kRowsPerWave;
    const int grid = (int)((nrows + rows_per_wg - 1) / rows_per_wg);
    hipLaunchKernelGGL((k_wave_rows<LEVELS, CHUNKS>), dim3(grid), dim3(64 * waves), waves * L::bytes, s,
                       Arow, Acol, Brow, Bcol, topw, rows, nrows, row_begin, Fprefix, tmp, cnt);
}

template <int LEVELS>
static void launch_levels(int bin, const int *Arow, const int *Acol, const int *Brow, const int *Bcol,
                          int topw, const int *rows, int nrows, int row_begin,
                          const long long *Fprefix, int *tmp, int *cnt, hipStream_t s)
{
    switch (bin) {
    case 1: launch_one<LEVELS, 1>(Arow, Acol, Brow, Bcol, topw, rows, nrows, row_begin, Fprefix, tmp, cnt, s); break;
    case 2: launch_one<LEVELS, 2>(Arow, Acol, Brow, Bcol, topw, rows, nrows, row_begin, Fprefix, tmp, cnt, s); break;
    case 3: launch_one<LEVELS, 4>(Arow, Acol, Brow, Bcol, topw, rows, nrows, row_begin, Fprefix, tmp, cnt, s); break;
    case 4: launch_one<LEVELS, 8>(Arow, Acol, Brow, Bcol, topw, rows, nrows, row_begin, Fprefix, tmp, cnt, s); break;
    case 5: launch_one<LEVELS, 16>(Arow, Acol, Brow, Bcol, topw, rows, nrows, row_begin, Fprefix, tmp, cnt, s); break;
    case 6: launch_one<LEVELS, 32>(Arow, Acol, Brow, Bcol, topw, rows, nrows, row_begin, Fprefix, tmp, cnt, s); break;
    default: break;
    }
}

void launch_wave_rows(int bin, int levels, const int *Arow, const int *Acol, const int *Brow,
                      const int *Bcol, int cols, const int *rows, int nrows, int row_begin,
                      const long long *Fprefix, int *tmp, int *cnt, hipStream_t s)
{
    if (nrows <= 0) return;
    // words of the directly addressed top bitmap: ceil(cols / 64^levels)
    const long long span = 1ll << (6 * levels);
    const int topw = (int)(((long long)cols + span - 1) / span);
    switch (levels) {
    case 1: launch_levels<1>(bin, Arow, Acol, Brow, Bcol, topw, rows, nrows, row_begin, Fprefix, tmp, cnt, s); break;
    case 2: launch_levels<2>(bin, Arow, Acol, Brow, Bcol, topw, rows, nrows, row_begin, Fprefix, tmp, cnt, s); break;
    case 3: launch_levels<3>(bin, Arow, Acol, Brow, Bcol, topw, rows, nrows, row_begin, Fprefix, tmp, cnt, s); break;
    default: launch_levels<4>(bin, Arow, Acol, Brow, Bcol, topw, rows, nrows, row_begin, Fprefix, tmp, cnt, s); break;
    }
}

}  // namespace bsp
