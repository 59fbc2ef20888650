// wave_masked.hip -- masked product C = F .* (A*B), one wavefront per row, MASK-FIRST.
//
// The reference (SpGEMM_masked, final/SpGEMM_mpi_omp.c:232-288) presets its flag array so that
// only the columns of F's row can ever be appended (:253-255), then runs the ordinary Gustavson
// loop.  The same idea with the rank bitmap of wave_rows.inc: the structure is built from the
// MASK row's columns (its size follows |F_i|, not the product count), all its levels are kept,
// and the row's products are then STREAMED through it read-only -- a product whose digit path
// exists down to a set level-0 bit marks that bit in a "kept" word.  Products never have to be
// resident, so a row may have any number of them; what limits the one-wave path is the mask row
// (<= 2048 entries, cols <= 2^23 so that three 5-bit levels suffice and no slot buffer is reused).
// Longer mask rows / wider matrices take the two-bitmap window kernel (dense_rows.hip).
//
// Products are walked 64 A-nonzeros at a time in windows of 256 products, with the same
// starts-bitmap flattening as the unmasked kernel (all 64 lanes load B.col_idx every step).
// Output: the kept bits are read out in rank order = ascending columns, staged in LDS, streamed
// to tmp[recpre ..) coalesced; cnt[row] = their number (<= |F_i|).
#include "kernels.hpp"
#include "wave.hpp"

namespace bsp {

constexpr int kMaskWinChunks = 4;       // products per window = 256, kept in registers
constexpr long long kMaskWaveMaxProducts = 8192;   // beyond this a row is streamed by a whole workgroup

template <int LEVELS, int CHUNKS>
struct MaskCfg {
    static constexpr int CAP = 64 * CHUNKS;
    static constexpr int TOPW = 256;
    static constexpr int bytes_per_wave = 4 * TOPW + 2 * TOPW + 8 * kMaskWinChunks + 4 * 64      // top, topPre, starts, delta
                                          + 4 * CAP * 3                                          // SA, K0, L0w
                                          + (LEVELS >= 3 ? 4 * CAP + 2 * CAP : 32)               // SB, preB
                                          + 0;
    static constexpr int w4 = (4 * bytes_per_wave > 64 * 1024) ? 0 : (160 * 1024 / (4 * bytes_per_wave)) * 4;
    static constexpr int w2 = (2 * bytes_per_wave > 64 * 1024) ? 0 : (160 * 1024 / (2 * bytes_per_wave)) * 2;
    static constexpr int WAVES = (w4 >= w2 && w4 > 0) ? 4 : (w2 > 0 ? 2 : 1);
};

template <int LEVELS, int CHUNKS>
__global__ __launch_bounds__((64 * MaskCfg<LEVELS, CHUNKS>::WAVES))
void k_wave_masked(const int2 *__restrict__ ab, const int *__restrict__ Bcol, int topw,
                   const int *__restrict__ Frow, const int *__restrict__ Fcol,
                   const RowRec *__restrict__ rec, const long long *__restrict__ recpre,
                   int nrows, int row_begin, int *__restrict__ tmp, int *__restrict__ cnt)
{
    using Cfg = MaskCfg<LEVELS, CHUNKS>;
    constexpr int CAP = Cfg::CAP, TOPW = Cfg::TOPW, WAVES = Cfg::WAVES, TW = TOPW / 64;
    constexpr int PCH = kMaskWinChunks;
    __shared__ __attribute__((aligned(16))) u32 s_top[WAVES][TOPW];
    __shared__ __attribute__((aligned(16))) unsigned short s_topPre[WAVES][TOPW];
    __shared__ __attribute__((aligned(16))) u64 s_starts[WAVES][PCH];
    __shared__ __attribute__((aligned(16))) int s_delta[WAVES][64];
    __shared__ __attribute__((aligned(16))) u32 s_SA[WAVES][CAP];           // level-0 masks of the MASK row
    constexpr int KN = (LEVELS == 1) ? TOPW : CAP;                 // level-0 slots: the top words themselves when LEVELS == 1
    __shared__ __attribute__((aligned(16))) u32 s_K0[WAVES][KN];            // kept bits
    __shared__ __attribute__((aligned(16))) u32 s_L0w[WAVES][CAP];          // word id of every level-0 slot; emit staging
    __shared__ __attribute__((aligned(16))) u32 s_SB[WAVES][LEVELS >= 3 ? CAP : 4];
    __shared__ __attribute__((aligned(16))) unsigned short s_preB[WAVES][LEVELS >= 3 ? CAP : 8];

    const int lane = lane_id();
    const int wave_in_wg = threadIdx.x >> 6;
    const long long wave_global = (long long)blockIdx.x * WAVES + wave_in_wg;
    const long long k0 = wave_global * kRowsPerWave;
    if (k0 >= nrows) return;                                       // wave-uniform; no barriers used
    const int nmine = (nrows - k0 < kRowsPerWave) ? (int)(nrows - k0) : kRowsPerWave;

    int r_row = 0, r_a0 = 0, r_alen = 0;
    long long r_pre = 0;
    if (lane < nmine) {
        const RowRec q = rec[k0 + lane];
        r_row = q.row;
        r_a0 = q.a0;
        r_alen = q.alen;
        r_pre = recpre[k0 + lane];
    }

    u32 *top = s_top[wave_in_wg];
    unsigned short *topPre = s_topPre[wave_in_wg];
    u64 *starts = s_starts[wave_in_wg];
    int *delta = s_delta[wave_in_wg];
    u32 *SA = s_SA[wave_in_wg], *K0 = s_K0[wave_in_wg], *L0w = s_L0w[wave_in_wg], *SB = s_SB[wave_in_wg];
    unsigned short *preB = s_preB[wave_in_wg];

    clear_blocked<TW>(top, lane);
    if (lane < PCH) starts[lane] = 0ull;
    clear_blocked<CHUNKS>(SA, lane);
    clear_blocked<KN / 64>(K0, lane);
    if (LEVELS >= 3) clear_blocked<CHUNKS>(SB, lane);
    wave_lds_fence();

    for (int kk = 0; kk < nmine; kk++) {
        const int i = wave_bcast(r_row, kk);
        const int a0 = wave_bcast(r_a0, kk);
        const int alen = wave_bcast(r_alen, kk);
        const u32 pre_lo = (u32)wave_bcast((int)(u32)r_pre, kk);
        const u32 pre_hi = (u32)wave_bcast((int)(u32)((unsigned long long)r_pre >> 32), kk);
        int *out = tmp + (long long)(((u64)pre_hi << 32) | pre_lo);
        const int f0 = Frow[i], mlen = Frow[i + 1] - f0;           // <= CAP by the row's class

        // ---- 1. rank bitmap of the mask row (all levels stay alive) ----------------------
        int mcol[CHUNKS], rank[CHUNKS];
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            const int p = c * 64 + lane;
            mcol[c] = Fcol[f0 + (p < mlen ? p : 0)];
        }
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            const bool ok = c * 64 + lane < mlen;
            const u32 cc = ok ? (u32)mcol[c] : 0u;
            mcol[c] = (int)cc;
            const u32 tw = cc >> (5 * LEVELS);
            if (ok) atomicOr(&top[tw], 1u << ((cc >> (5 * (LEVELS - 1))) & 31));   // tail lanes masked off
            rank[c] = (int)tw;
        }
        wave_lds_fence();
        u32 *L0 = top;                      // level-0 masks of the mask row
        if (LEVELS >= 2) {
            scan_blocked<TW>(top, topPre, lane);
            wave_lds_fence();
            const u32 *P = top;
            const unsigned short *Ppre = topPre;
#pragma unroll
            for (int lev = LEVELS - 2; lev >= 0; lev--) {
                u32 *S = (lev == 1) ? SB : SA;
                unsigned short *Spre = preB;                       // only level 1 needs ranks
#pragma unroll
                for (int c = 0; c < CHUNKS; c++) {
                    const bool ok = c * 64 + lane < mlen;
                    const u32 cc = (u32)mcol[c];
                    const u32 x = P[rank[c]];
                    const int pre = Ppre[rank[c]];
                    const u32 b = (cc >> (5 * (lev + 1))) & 31;
                    const int r2 = pre + __popc(x & ((1u << b) - 1u));
                    if (ok) {
                        atomicOr(&S[r2], 1u << ((cc >> (5 * lev)) & 31));
                        if (lev == 0) L0w[r2] = cc >> 5;
                    }
                    rank[c] = r2;
                }
                wave_lds_fence();
                if (lev > 0) {
                    scan_blocked<CHUNKS>(S, Spre, lane);
                    wave_lds_fence();
                    P = S;
                    Ppre = Spre;
                }
            }
            L0 = SA;
        }

        // ---- 2. stream the row's products through the structure, read-only ---------------
        for (int ab0 = 0; ab0 < alen; ab0 += 64) {
            int2 e = make_int2(0, 0);
            if (ab0 + lane < alen) e = ab[a0 + ab0 + lane];
            const int bs = e.x, len = e.y;
            const int inc = wave_incl_scan(len);
            const int excl = inc - len;
            const int Fb = wave_bcast(inc, 63);                    // products of this batch of 64 sources
            for (int w0 = 0; w0 < Fb; w0 += 64 * PCH) {
                // sources that own products inside the window [w0, w0 + 256)
                const bool part = len > 0 && excl < w0 + 64 * PCH && excl + len > w0;
                const u64 bal = __ballot(part);
                if (part) {
                    const int sidx = __popcll(bal & mask_lt(lane));
                    const int pos = (excl > w0 ? excl : w0) - w0;
                    delta[sidx] = bs - excl;                       // B address = delta + batch product index
                    atomicOr(&starts[pos >> 6], 1ull << (pos & 63));
                }
                wave_lds_fence();
                u64 sw = 0ull;
                if (lane < PCH) { sw = starts[lane]; starts[lane] = 0ull; }
                const int sinc = wave_incl_scan(__popcll(sw));
                const int sbefore = sinc - __popcll(sw);
                int gaddr[PCH];
#pragma unroll
                for (int c = 0; c < PCH; c++) {
                    const int p = c * 64 + lane;
                    const u64 M = wave_bcast64(sw, c);
                    const int before = wave_bcast(sbefore, c);
                    const bool ok = w0 + p < Fb;
                    int s = before + __popcll(M & mask_le(lane)) - 1;
                    s = ok ? s : 0;
                    gaddr[c] = ok ? delta[s] + w0 + p : 0;          // tail lanes: Bcol[0]
                }
                int pc[PCH];
#pragma unroll
                for (int c = 0; c < PCH; c++) {
                    const bool ok = w0 + c * 64 + lane < Fb;
                    pc[c] = ok ? Bcol[gaddr[c]] : -1;
                }
                wave_lds_fence();
                // probe: follow the digit path; mark the kept bit when every level has it
#pragma unroll
                for (int c = 0; c < PCH; c++) {
                    const bool ok = pc[c] >= 0;
                    const u32 cc = ok ? (u32)pc[c] : 0u;
                    const u32 tw = cc >> (5 * LEVELS);
                    bool hit = ok && tw < (u32)topw;
                    u32 x = top[hit ? tw : 0];
                    int r = (int)(hit ? tw : 0);
                    if (LEVELS >= 2) {
                        const u32 b = (cc >> (5 * (LEVELS - 1))) & 31;
                        hit = hit && ((x >> b) & 1u);
                        r = topPre[r] + __popc(x & ((1u << b) - 1u));
                        r = hit ? r : 0;
                        if (LEVELS >= 3) {
                            x = SB[r];
                            const u32 b1 = (cc >> 5) & 31;
                            hit = hit && ((x >> b1) & 1u);
                            r = preB[r] + __popc(x & ((1u << b1) - 1u));
                            r = hit ? r : 0;
                        }
                        x = SA[r];
                    }
                    const u32 b0 = cc & 31;
                    hit = hit && ((x >> b0) & 1u);
                    // only the hits touch K0: parking the misses (the vast majority of a sparse
                    // masked product) on one spare word serialises them as same-address atomics
                    if (hit) atomicOr(&K0[r], 1u << b0);
                }
                wave_lds_fence();
            }
        }

        // ---- 3. emit the kept bits in rank order; clear everything for the next row ------
        constexpr int W0 = (LEVELS == 1) ? TW : CHUNKS;
        u32 m[W0], wv[W0];
#pragma unroll
        for (int k = 0; k < W0; k++) {
            m[k] = K0[lane * W0 + k];
            wv[k] = (LEVELS == 1) ? (u32)(lane * W0 + k) : L0w[lane * W0 + k];
        }
        clear_blocked<W0>(K0, lane);
        clear_blocked<W0>(L0, lane);
        if (LEVELS >= 2) clear_blocked<TW>(top, lane);
        if (LEVELS >= 3) clear_blocked<CHUNKS>(SB, lane);
        int mine = 0;
#pragma unroll
        for (int k = 0; k < W0; k++) mine += __popc(m[k]);
        const int inc = wave_incl_scan(mine);
        const int running = wave_bcast(inc, 63);
        wave_lds_fence();
        {
            int pos = inc - mine;
#pragma unroll
            for (int k = 0; k < W0; k++) {
                u32 mk = m[k];
                const u32 base = wv[k] << 5;
                while (mk) {
                    L0w[stage_swz(pos)] = base | (u32)__builtin_ctz(mk);
                    pos++;
                    mk &= mk - 1u;
                }
            }
        }
        wave_lds_fence();
        for (int t = lane; t < running; t += 64) __builtin_nontemporal_store((int)L0w[stage_swz(t)], out + t);   // streamed, as in wave_rows.inc
        if (lane == 0) cnt[i - row_begin] = running;
        wave_lds_fence();
    }
}

template <int LEVELS, int CHUNKS>
static void launch_mask_one(const int2 *ab, const int *Bcol, int topw, const int *Frow, const int *Fcol,
                            const RowRec *rec, const long long *recpre, int nrows, int row_begin,
                            int *tmp, int *cnt, hipStream_t s)
{
    using Cfg = MaskCfg<LEVELS, CHUNKS>;
    const long long rows_per_wg = (long long)Cfg::WAVES * kRowsPerWave;
    const int grid = (int)((nrows + rows_per_wg - 1) / rows_per_wg);
    hipLaunchKernelGGL((k_wave_masked<LEVELS, CHUNKS>), dim3(grid), dim3(64 * Cfg::WAVES), 0, s,
                       ab, Bcol, topw, Frow, Fcol, rec, recpre, nrows, row_begin, tmp, cnt);
}

template <int LEVELS>
static void launch_mask_levels(int bin, const int2 *ab, const int *Bcol, int topw, const int *Frow,
                               const int *Fcol, const RowRec *rec, const long long *recpre, int nrows,
                               int row_begin, int *tmp, int *cnt, hipStream_t s)
{
    // the mask-first kernel is instantiated for 7 mask-row capacities; a class uses the smallest
    // one that holds its rows (mask rows are short: the fine classes of the plain product buy nothing)
    const int chunks = kWaveChunks[bin];
#define BSP_MASK(C) launch_mask_one<LEVELS, C>(ab, Bcol, topw, Frow, Fcol, rec, recpre, nrows, row_begin, tmp, cnt, s)
    if (chunks <= 1) BSP_MASK(1);
    else if (chunks <= 2) BSP_MASK(2);
    else if (chunks <= 4) BSP_MASK(4);
    else if (chunks <= 8) BSP_MASK(8);
    else if (chunks <= 12) BSP_MASK(12);
    else if (chunks <= 16) BSP_MASK(16);
    else BSP_MASK(32);
#undef BSP_MASK
}

bool wave_masked_supported(int cols) { return levels_for_cols(cols) <= 3; }

void launch_wave_masked(int bin, const int2 *ab, const int *Bcol, int cols, const int *Frow, const int *Fcol,
                        const RowRec *rec, const long long *recpre, int nrows, int row_begin,
                        int *tmp, int *cnt, hipStream_t s)
{
    if (nrows <= 0) return;
    const int levels = levels_for_cols(cols);
    const long long span = 1ll << (5 * levels);
    const int topw = (int)(((long long)cols + span - 1) / span);
    switch (levels) {
    case 1: launch_mask_levels<1>(bin, ab, Bcol, topw, Frow, Fcol, rec, recpre, nrows, row_begin, tmp, cnt, s); break;
    case 2: launch_mask_levels<2>(bin, ab, Bcol, topw, Frow, Fcol, rec, recpre, nrows, row_begin, tmp, cnt, s); break;
    default: launch_mask_levels<3>(bin, ab, Bcol, topw, Frow, Fcol, rec, recpre, nrows, row_begin, tmp, cnt, s); break;
    }
}

// mask length per row (0 when the row has no products): what the masked multiply bins and offsets by
__global__ void k_mask_lengths(const long long *__restrict__ F, const int *__restrict__ Frow, int row_begin,
                               int n, long long *__restrict__ mlen)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        long long m = F[i] > 0 ? (long long)(Frow[row_begin + i + 1] - Frow[row_begin + i]) : 0;
        // a wave streams its row's products 256 at a time: rows with very many products go to the
        // 1024-thread dense kernel whatever their mask length (class = dense when m > kMaxWaveCap;
        // the larger m only reserves more room in tmp)
        if (m > 0 && m <= kMaxWaveCap && F[i] > kMaskWaveMaxProducts) m = kMaxWaveCap + 1;
        mlen[i] = m;
    }
}

void launch_mask_lengths(const long long *F, const int *Frow, int row_begin, int n, long long *mlen, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_mask_lengths, dim3((n + 255) / 256), dim3(256), 0, s, F, Frow, row_begin, n, mlen);
}

}  // namespace bsp
