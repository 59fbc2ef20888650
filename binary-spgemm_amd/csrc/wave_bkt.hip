// wave_bkt.hip -- the BUCKET accumulator of the one-wave-per-row numeric pass (round 4).
//
// Same job as k_wave_rows (wave_rows.inc): the products of one A-row -- the B rows its nonzeros select, gathered into
// registers -- become the sorted, duplicate-free row of C.  Replaces SpGEMM_bigslice's probe / append / quickSort / reset
// of one row (final/SpGEMM_mpi_omp.c:33-50, final/utils.c:159-173).
//
// The rank bitmap of wave_rows.inc pays, per product, SEVEN random LDS accesses (three ranked sweeps), and a random
// 4-byte access of 32 lanes over 32 banks costs ~3.5 conflict cycles: 52 % of that kernel's LDS-array cycles are bank
// conflicts (profiles/r03_pmc_sq_summary.txt) and the alignment probe of round 4 (profiles/r04_alignment_probe.log) shows
// it is not waiting for memory.  Here a product makes THREE:
//     1. one returning ds_add on the counter of its RANGE bucket  b = (col - row_min) >> shift      -> arrival number j
//     2. after a blocked scan of the counters (exclusive bases, in place): one read of base[b]
//     3. one write of the column to sorted[base[b] + j]
// Buckets are ranges, so `sorted` is the row in order up to the order INSIDE each bucket.  The row goes back into
// registers blocked (lane l holds SW consecutive entries) and odd-even transposition phases -- plain v_min/v_max on
// registers, one DPP pair across lanes -- put every bucket in order: a pair from two buckets never swaps, a bucket of m
// entries is sorted after m phases, so ceil(mc / 2) double phases suffice where mc is the row's fullest bucket (known
// from the scan; 3-5 at load <= 1).  Duplicates are neighbours afterwards; a row without any (98 % on the bench
// matrix) is stored straight from the registers.
//
// A row whose fullest bucket exceeds kBktMaxLoad (clustered columns, heavy duplicates: range buckets cannot split them)
// is NOT processed here: its record goes to a device-side list and the rank-bitmap kernel takes it in list mode right
// after this launch, on the same stream (launch_wave_rows, nrows_dev).  Independent of the column count: no LEVELS.
//
// Roofline: HBM (gather of B.col_idx, 4 B per product, + 4 B per output written).  No MFMA.
#include "kernels.hpp"
#include "wave.hpp"

namespace bsp {

constexpr int kBktMaxLoad = 16;
constexpr int bkt_floor_log2(int x) { int k = 0; while ((2 << k) <= x) k++; return k; }
constexpr int bkt_prev_chunks(int chunks)
{
    int prev = 0;
    for (int b = 1; b <= kWaveBins; b++) {
        if (kWaveChunks[b] == chunks) return prev;
        prev = kWaveChunks[b];
    }
    return 0;
}

template <int CHUNKS>
struct BktCfg {
    static constexpr int CAP = 64 * CHUNKS;
    static constexpr int SW = (CHUNKS <= 2) ? CHUNKS : ((CHUNKS + 3) & ~3);   // blocked entries per lane (whole 16-byte vectors)
    static constexpr int SLOTS = 64 * SW;
    static constexpr int NBW = SW <= 2 ? 2 : (SW <= 4 ? 4 : (SW <= 8 ? 8 : (SW <= 16 ? 16 : 32)));   // counters per lane, a power of two: 64 * NBW buckets (load <= 1)
    static constexpr int NB = 64 * NBW;
    static constexpr int NBK = bkt_floor_log2(NB);                 // buckets in use: 2^NBK
    static constexpr int WAVES = 4;
    static constexpr int RPW = (CHUNKS >= 16) ? kRowsPerWave / 2 : kRowsPerWave;
    static constexpr int FULL = bkt_prev_chunks(CHUNKS);           // chunks that are full for every row of the class
};

template <int CHUNKS>
__global__ __launch_bounds__((64 * BktCfg<CHUNKS>::WAVES))
void k_wave_bkt(const int2 *__restrict__ ab, const int *__restrict__ Bcol,
                const RowRec *__restrict__ rec, const long long *__restrict__ recpre,
                const long long *__restrict__ row_ptr, int nrows, int rpw, int row_begin,
                int *__restrict__ tmp, int *__restrict__ cnt,
                RowRec *__restrict__ fb_rec, long long *__restrict__ fb_pre, int *__restrict__ fb_count,
                unsigned *__restrict__ err)
{
    using Cfg = BktCfg<CHUNKS>;
    constexpr int WAVES = Cfg::WAVES, SW = Cfg::SW, SLOTS = Cfg::SLOTS, FULL = Cfg::FULL, NBW = Cfg::NBW, NBK = Cfg::NBK;
    __shared__ __attribute__((aligned(16))) u64 s_starts[WAVES][CHUNKS];
    __shared__ __attribute__((aligned(16))) u32 s_row[WAVES][SLOTS < 64 * 2 ? 64 * 2 : SLOTS];   // delta[] of the gather, then the scattered row, then staging
    __shared__ __attribute__((aligned(16))) u32 s_cw[WAVES][Cfg::NB];                              // bucket counters / bases

    const int lane = lane_id();
    const int plane = (int)(__brev((unsigned)lane) >> 26);         // (as in wave_rows.inc: neighbouring lanes, different B rows)
    const int wave_in_wg = threadIdx.x >> 6;
    const long long k0 = ((long long)blockIdx.x * WAVES + wave_in_wg) * rpw;
    if (k0 >= nrows) return;                                       // wave-uniform; no barriers used
    const int nmine = (nrows - k0 < rpw) ? (int)(nrows - k0) : rpw;

    int r_row = 0, r_a0 = 0, r_alen = 0, r_f = 0;
    long long r_pre = 0;
    if (lane < nmine) {
        const RowRec q = rec[k0 + lane];
        r_row = q.row;
        r_a0 = q.a0;
        r_alen = q.alen;
        r_f = q.f;
        r_pre = row_ptr ? row_ptr[q.row - row_begin] : recpre[k0 + lane];
    }
    u64 *starts = s_starts[wave_in_wg];
    u32 *rowbuf = s_row[wave_in_wg];
    int *delta = reinterpret_cast<int *>(rowbuf);
    u32 *cw = s_cw[wave_in_wg];
    if (lane < CHUNKS) starts[lane] = 0ull;
    clear_blocked<NBW>(cw, lane);
    wave_lds_fence();

    int2 ab_next = make_int2(0, 0);
    {
        const int a0 = wave_bcast(r_a0, 0), alen = wave_bcast(r_alen, 0);
        if (lane < alen) ab_next = ab[a0 + lane];
    }
    int my_cnt = -1;                                               // -1: the row was handed to the list
    for (int kk = 0; kk < nmine; kk++) {
        const int a0 = wave_bcast(r_a0, kk);
        const int alen = wave_bcast(r_alen, kk);
        const u32 pre_lo = (u32)wave_bcast((int)(u32)r_pre, kk);
        const u32 pre_hi = (u32)wave_bcast((int)(u32)((unsigned long long)r_pre >> 32), kk);
        const long long out_off = (long long)(((u64)pre_hi << 32) | pre_lo);
        int *out = tmp + out_off;

        // ---- gather plan (wave_rows.inc): product offsets of the selected B rows ----------------
        int F = 0, nsrc = 0;
        for (int ab0 = 0; ab0 < alen; ab0 += 64) {
            int2 e = ab_next;
            if (ab0 > 0) {
                e = make_int2(0, 0);
                if (ab0 + lane < alen) e = ab[a0 + ab0 + lane];
            }
            const int bs = e.x, len = e.y;
            const int inc = wave_incl_scan(len);
            const int excl = F + inc - len;
            const u64 bal = __ballot(len > 0);
            if (len > 0 && (unsigned)excl < (unsigned)Cfg::CAP) {   // (the bound only trips on inconsistent operands)
                const int sidx = nsrc + __popcll(bal & mask_lt(lane));
                delta[sidx] = bs - excl;
                atomicOr(&starts[excl >> 6], 1ull << (excl & 63));
            }
            F += wave_bcast(inc, 63);
            nsrc += __popcll(bal);
        }
        if (F > Cfg::CAP || F < 0) {
            if (lane == 0) atomicOr(err, kErrCapacity);
            F = F < 0 ? 0 : Cfg::CAP;
        }
        wave_lds_fence();
        u64 sw = 0ull;
        if (lane < CHUNKS) { sw = starts[lane]; starts[lane] = 0ull; }
        const int sinc = wave_incl_scan(__popcll(sw));
        const int sbefore = sinc - __popcll(sw);

        // ---- gather B.col_idx: all lanes busy, products kept in registers -------------------------
        u32 col[CHUNKS];
        int gaddr[CHUNKS];
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            const int p = c * 64 + plane;
            const u64 M = wave_bcast64(sw, c);
            const int before = wave_bcast(sbefore, c);
            int s = before + __popcll(M & mask_le(plane)) - 1;
            s = (c < FULL || p < F) ? s : 0;
            gaddr[c] = delta[s];
        }
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            const int p = c * 64 + plane;
            const int g = gaddr[c] + ((c < FULL || p < F) ? p : 0);
            col[c] = (u32)Bcol[g];
        }
        ab_next = make_int2(0, 0);
        if (kk + 1 < nmine) {
            const int na0 = wave_bcast(r_a0, kk + 1), nalen = wave_bcast(r_alen, kk + 1);
            if (lane < nalen) ab_next = ab[na0 + lane];
        }
        wave_lds_fence();   // delta is dead from here on

        // ---- buckets: the row's column range cut into 2^NBK equal pieces -------------------------
        u32 mn = ~0u, mx = 0u;
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            const bool ok = c < FULL || c * 64 + plane < F;
            mn = ok ? min(mn, col[c]) : mn;
            mx = ok ? max(mx, col[c]) : mx;
        }
        mn = wave_min_u32(mn);
        mx = wave_max_u32(mx);
        const u32 span = F > 0 ? mx - mn : 0u;                     // column range - 1
        int sh = 32 - __builtin_clz(span | 1u) - NBK;              // (span >> sh) < 2^NBK
        sh = sh < 0 ? 0 : sh;
        u32 jb[CHUNKS];                                            // bucket << 16 | arrival number in the bucket
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            const bool ok = c < FULL || c * 64 + plane < F;
            const u32 b = (col[c] - mn) >> sh;
            jb[c] = b << 16;
            if (ok) jb[c] |= atomicAdd(&cw[b], 1u);
        }
        wave_lds_fence();
        // counts -> exclusive bases, in place (lane l owns NBW consecutive counters); the fullest bucket
        u32 mc = 0u;
        {
            u32 w[NBW];
#pragma unroll
            for (int k = 0; k < NBW; k++) w[k] = cw[lane * NBW + k];
            u32 run = 0u;
#pragma unroll
            for (int k = 0; k < NBW; k++) {
                const u32 c0 = w[k];
                mc = max(mc, c0);
                w[k] = run;
                run += c0;
            }
            const u32 base = (u32)wave_incl_scan((int)run) - run;
#pragma unroll
            for (int k = 0; k < NBW; k++) cw[lane * NBW + k] = w[k] + base;
            mc = wave_max_u32(mc);
        }
        wave_lds_fence();
        if (mc > (u32)kBktMaxLoad) {
            // clustered row: hand it to the rank-bitmap kernel (list mode), nothing emitted here
            if (lane == 0) {
                const int slot = atomicAdd(fb_count, 1);
                RowRec q;
                q.row = wave_bcast(r_row, kk);
                q.a0 = a0;
                q.alen = alen;
                q.f = wave_bcast(r_f, kk);
                fb_rec[slot] = q;
                fb_pre[slot] = out_off;
            }
            clear_blocked<NBW>(cw, lane);
            wave_lds_fence();
            continue;                                              // (my_cnt of lane kk stays -1)
        }
        // ---- scatter into bucket order ----------------------------------------------------------------
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            const bool ok = c < FULL || c * 64 + plane < F;
            if (ok) rowbuf[cw[jb[c] >> 16] + (jb[c] & 0xffffu)] = col[c];
        }
        wave_lds_fence();
        clear_blocked<NBW>(cw, lane);                              // all zero for the next row
        // ---- the row back in registers, blocked: lane l holds entries [l * SW, l * SW + SW) -----------
        u32 x[SW];
#pragma unroll
        for (int k = 0; k < SW; k++) x[k] = rowbuf[lane * SW + k];
#pragma unroll
        for (int k = 0; k < SW; k++) x[k] = (lane * SW + k < F) ? x[k] : ~0u;     // beyond the row: sentinels above every column
        const int iters = mc <= 1u ? 0 : (int)(mc + 1u) >> 1;      // m >= 2 entries of a bucket: m phases; one iteration = two phases
        if constexpr (SW == 1) {
            for (int it = 0; it < iters; it++) {
                // even phase: lanes (2m, 2m+1); odd phase: lanes (2m+1, 2m+2)
                const u32 pr = dpp_or_old<0xB1, 0xF>(x[0], x[0]);  // quad_perm [1,0,3,2]: the other lane of the pair
                x[0] = (lane & 1) ? max(x[0], pr) : min(x[0], pr);
                const u32 nf = wave_next(~0u, x[0]), pl = wave_prev(0u, x[0]);
                x[0] = (lane & 1) ? min(x[0], nf) : max(x[0], pl);
            }
        } else {
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int k = 0; k + 1 < SW; k += 2) {
                    const u32 lo = min(x[k], x[k + 1]), hi = max(x[k], x[k + 1]);
                    x[k] = lo;
                    x[k + 1] = hi;
                }
#pragma unroll
                for (int k = 1; k + 1 < SW; k += 2) {
                    const u32 lo = min(x[k], x[k + 1]), hi = max(x[k], x[k + 1]);
                    x[k] = lo;
                    x[k + 1] = hi;
                }
                const u32 nf = wave_next(~0u, x[0]), pl = wave_prev(0u, x[SW - 1]);
                x[SW - 1] = min(x[SW - 1], nf);
                x[0] = max(x[0], pl);
            }
        }
        // ---- duplicates are neighbours now: count them (sentinels have the top bit set) ---------------
        int dups = 0;
        {
            u32 pv = wave_prev(~0u, x[SW - 1]);
#pragma unroll
            for (int k = 0; k < SW; k++) {
                dups += __popcll(__ballot(x[k] == pv && (int)x[k] >= 0));
                pv = x[k];
            }
        }
        const int running = F - dups;                              // |C_i|
        if (dups == 0) {
            // the registers are the row: lane l stores its SW consecutive entries, clipped by the descriptor; streamed (nt)
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)out, (short)0, F * 4, 0x00020000);
            if constexpr (SW % 4 == 0) {
                typedef u32 v4u __attribute__((ext_vector_type(4)));
#pragma unroll
                for (int k = 0; k < SW; k += 4) {
                    const v4u v = {x[k], x[k + 1], x[k + 2], x[k + 3]};
                    __builtin_amdgcn_raw_buffer_store_b128(v, rs, (lane * SW + k) * 4, 0, 2);
                }
            } else if constexpr (SW == 2) {
                typedef u32 v2u __attribute__((ext_vector_type(2)));
                const v2u v = {x[0], x[1]};
                __builtin_amdgcn_raw_buffer_store_b64(v, rs, lane * 8, 0, 2);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(x[0], rs, lane * 4, 0, 2);
            }
        } else {
            // squeeze the kept entries together through the staging row (every lane has read its entries), store coalesced
            u32 keepm = 0u;
            {
                u32 pv = wave_prev(~0u, x[SW - 1]);
#pragma unroll
                for (int k = 0; k < SW; k++) {
                    keepm |= ((int)x[k] >= 0 && x[k] != pv) ? (1u << k) : 0u;
                    pv = x[k];
                }
            }
            const int mine = __popc(keepm);
            int pos = wave_incl_scan(mine) - mine;
            wave_lds_fence();
#pragma unroll
            for (int k = 0; k < SW; k++)
                if (keepm & (1u << k)) {
                    rowbuf[stage_swz(pos)] = x[k];
                    pos++;
                }
            wave_lds_fence();
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)out, (short)0, running * 4, 0x00020000);
#pragma unroll
            for (int c = 0; c < CHUNKS; c++) {
                const int t = c * 64 + lane;
                __builtin_amdgcn_raw_buffer_store_b32(rowbuf[stage_swz(t)], rs, t * 4, 0, 2);
            }
        }
        my_cnt = (lane == kk) ? running : my_cnt;
        wave_lds_fence();
    }
    if (cnt && lane < nmine && my_cnt >= 0) cnt[r_row - row_begin] = my_cnt;
}

template <int CHUNKS>
static void launch_bkt(const int2 *ab, const int *Bcol, const RowRec *rec, const long long *recpre, const long long *row_ptr,
                       int nrows, int row_begin, int *tmp, int *cnt, RowRec *fb_rec, long long *fb_pre, int *fb_count,
                       unsigned *err, hipStream_t s)
{
    using Cfg = BktCfg<CHUNKS>;
    constexpr int kSpreadWaves = 256 * 8;
    int rpw = (int)((nrows + kSpreadWaves - 1) / kSpreadWaves);
    if (rpw > Cfg::RPW) rpw = Cfg::RPW;
    if (rpw < 1) rpw = 1;
    const long long rows_per_wg = (long long)Cfg::WAVES * rpw;
    const int grid = (int)((nrows + rows_per_wg - 1) / rows_per_wg);
    hipLaunchKernelGGL((k_wave_bkt<CHUNKS>), dim3(grid), dim3(64 * Cfg::WAVES), 0, s, ab, Bcol, rec, recpre, row_ptr, nrows, rpw,
                       row_begin, tmp, cnt, fb_rec, fb_pre, fb_count, err);
}

void launch_wave_bkt(int bin, const int2 *ab, const int *Bcol, const RowRec *rec, const long long *recpre,
                     const long long *row_ptr, int nrows, int row_begin, int *tmp, int *cnt,
                     RowRec *fb_rec, long long *fb_pre, int *fb_count, unsigned *err, hipStream_t s)
{
    if (nrows <= 0) return;
    switch (bin) {
#define BSP_CASE(b) case b: launch_bkt<kWaveChunks[b]>(ab, Bcol, rec, recpre, row_ptr, nrows, row_begin, tmp, cnt, fb_rec, fb_pre, fb_count, err, s); break;
    BSP_CASE(1) BSP_CASE(2) BSP_CASE(3) BSP_CASE(4) BSP_CASE(5) BSP_CASE(6) BSP_CASE(7) BSP_CASE(8)
    BSP_CASE(9) BSP_CASE(10) BSP_CASE(11) BSP_CASE(12) BSP_CASE(13) BSP_CASE(14) BSP_CASE(15) BSP_CASE(16)
#undef BSP_CASE
    static_assert(kWaveBins == 16, "one case per capacity class");
    default: break;
    }
}

}  // namespace bsp
