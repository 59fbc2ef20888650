// wave_count.hip -- the symbolic pass: exact |C_i| of every one-wave row BEFORE anything is emitted.
//
// The reference learns a row's size while it appends to it and grows Ccol with realloc
// (final/SpGEMM_mpi_omp.c:28-31, 38-42).  Here the sizes are obtained first, scanned into C.row_ptr,
// C.col_idx is allocated with exactly nnz(C) entries and the numeric pass (wave_rows.inc) writes
// every row at its final place -- no upper-bound placement, no compaction round trip.
//
// Counting distinct columns needs set semantics but no order, so this pass does not build the rank
// bitmap: the row's products go through an open-addressing hash set in LDS, one returning
// ds_cmpst_b32 per product (plus linear probing on a collision, table load <= 0.5):
//     slot EMPTY       -> the column is new, now stored
//     slot == column   -> a duplicate product (what `if (!xb[k])` skips at :38)
//     anything else    -> next slot
// |C_i| = F_i - duplicates.  About a fifth of the LDS work of the numeric pass per product; the
// gather (extents from the prepass, starts bitmap, all 64 lanes loading B.col_idx on every step)
// is the one of wave_rows.inc.  One template instance per capacity class, same record lists.
//
// Roofline: HBM/Infinity-Cache gather of B.col_idx, 4 B per product + 8 B per A-nonzero read,
// 4 B per row written.  No MFMA.
#include "kernels.hpp"
#include "wave.hpp"

namespace bsp {

constexpr int pow2_ge(int x) { int p = 1; while (p < x) p <<= 1; return p; }
constexpr int log2_of(int p) { int l = 0; while ((1 << l) < p) l++; return l; }

// hash slots per product of capacity, as a fraction (the table is the next power of two)
#define BSP_COUNT_NUM 3

template <int CHUNKS>
struct CountCfg {
    static constexpr int CAP = 64 * CHUNKS;
    // a power of two (the slot is the top bits of a multiplicative hash), load <= 1/3: about one
    // product in eight finds its first slot taken by another column
    static constexpr int H = pow2_ge(BSP_COUNT_NUM * CAP);
    static constexpr int LOGH = log2_of(H);
    static constexpr int bytes_per_wave = 4 * H + 4 * CAP + 8 * CHUNKS;      // table, delta (later: unsettled keys), starts
    // resident waves per CU (at most 32) with 4- or 2-wave workgroups; LDS is what limits them
    static constexpr int cap32(int w) { return w > 32 ? 32 : w; }
    static constexpr int w4 = (4 * bytes_per_wave > 64 * 1024) ? 0 : cap32((160 * 1024 / (4 * bytes_per_wave)) * 4);
    static constexpr int w2 = (2 * bytes_per_wave > 64 * 1024) ? 0 : cap32((160 * 1024 / (2 * bytes_per_wave)) * 2);
    static constexpr int WAVES = (w4 >= w2 && w4 > 0) ? 4 : (w2 > 0 ? 2 : 1);
    static constexpr int RPW = (CHUNKS >= 16) ? kRowsPerWave / 2 : kRowsPerWave;
};

constexpr u32 kEmptySlot = 0xffffffffu;      // never a column (columns are < 2^31)

template <int W>
__device__ __forceinline__ void fill_blocked(u32 *P, int lane, u32 v)
{
#pragma unroll
    for (int k = 0; k < W; k++) P[lane * W + k] = v;
}

// slot of a column: top LOGH bits of a multiplicative hash.  Columns below 2^24 (WIDE = false) take
// the full-rate 24-bit multiply; wider ones the 32-bit one.
template <int LOGH, bool WIDE>
__device__ __forceinline__ u32 hash_slot(u32 col)
{
    const u32 m = WIDE ? col * 0x9E3779B1u : __umul24(col, 0x9E3779u) * 1u;
    return m >> (32 - LOGH);
}

struct __attribute__((packed, aligned(4))) Int2U { int x, y; };   // 8 B, only dword aligned

template <int CHUNKS, bool WIDE>
__global__ __launch_bounds__((64 * CountCfg<CHUNKS>::WAVES))
void k_wave_count(const int *__restrict__ Acol, const int *__restrict__ Brow, const int *__restrict__ Bcol,
                  int2 *__restrict__ ab, const RowRec *__restrict__ rec, int nrows, int rpw, int row_begin,
                  int *__restrict__ cnt)
{
    using Cfg = CountCfg<CHUNKS>;
    constexpr int WAVES = Cfg::WAVES, H = Cfg::H, LOGH = Cfg::LOGH, CAP = Cfg::CAP;
    __shared__ __attribute__((aligned(16))) u32 s_tab[WAVES][H];
    __shared__ __attribute__((aligned(16))) int s_delta[WAVES][CAP];
    __shared__ __attribute__((aligned(16))) u64 s_starts[WAVES][CHUNKS];

    const int lane = lane_id();
    const int wave_in_wg = threadIdx.x >> 6;
    const long long wave_global = (long long)blockIdx.x * WAVES + wave_in_wg;
    const long long k0 = wave_global * rpw;
    if (k0 >= nrows) return;                                       // wave-uniform; no barriers used
    const int nmine = (nrows - k0 < rpw) ? (int)(nrows - k0) : rpw;

    int r_row = 0, r_a0 = 0, r_alen = 0;
    if (lane < nmine) {
        const RowRec q = rec[k0 + lane];
        r_row = q.row;
        r_a0 = q.a0;
        r_alen = q.alen;
    }
    u32 *tab = s_tab[wave_in_wg];
    int *delta = s_delta[wave_in_wg];
    u32 *unsettled = reinterpret_cast<u32 *>(s_delta[wave_in_wg]);   // delta is dead once the products are loaded
    u64 *starts = s_starts[wave_in_wg];
    fill_blocked<H / 64>(tab, lane, kEmptySlot);                   // once per wave: rows wipe what they touched
    if (lane < CHUNKS) starts[lane] = 0ull;
    wave_lds_fence();

    // Software pipeline over the rows of this wave, three rows deep.  While row k goes through the
    // hash set:  the B.col_idx gather of row k+1 is in flight (its plan -- extents, starts bitmap,
    // source offsets -- is made first; delta[] is free again once the gather addresses are in
    // registers);  the B.row_ptr pairs of row k+2's A-nonzeros are being gathered (one 8-byte random
    // access each: what the prepass kernel of the masked product spends its whole time on is hidden
    // here);  and the A.col_idx of row k+3 is being read.  The extents are left in ab[] for the
    // numeric pass, which then reads them coalesced.
    int acol_next = -1;                                            // A.col_idx of the row after next, one per lane
    int2 ab_next = make_int2(0, 0);                                // (B.row_ptr[j], |B_j|) of the next row
    auto load_acol = [&](int k) {
        acol_next = -1;
        if (k < nmine) {
            const int na0 = wave_bcast(r_a0, k), nalen = wave_bcast(r_alen, k);
            if (lane < nalen) acol_next = Acol[na0 + lane];
        }
    };
    auto extent_of = [&](int j) {
        int2 e = make_int2(0, 0);
        if (j >= 0) {
            const Int2U pr = *reinterpret_cast<const Int2U *>(Brow + j);   // one 8-B gather (dword aligned)
            e = make_int2(pr.x, pr.y - pr.x);
        }
        return e;
    };
    auto prefetch_extents = [&]() { ab_next = extent_of(acol_next); };
    u32 coln[CHUNKS];                                              // products of the NEXT row, in flight
    int Fn = 0;
    auto issue_gather = [&](int k) {                               // wave-uniform k < nmine
        const int a0 = wave_bcast(r_a0, k);
        const int alen = wave_bcast(r_alen, k);
        // ---- gather plan (see wave_rows.inc) ---------------------------------------------
        int F = 0, nsrc = 0;
        for (int ab0 = 0; ab0 < alen; ab0 += 64) {
            int2 e = ab_next;
            if (ab0 > 0)                                           // a row with more than 64 A-nonzeros (rare)
                e = extent_of(ab0 + lane < alen ? Acol[a0 + ab0 + lane] : -1);
            if (ab0 + lane < alen) ab[a0 + ab0 + lane] = e;        // for the numeric pass
            const int bs = e.x, len = e.y;
            const int inc = wave_incl_scan(len);
            const int excl = F + inc - len;
            const u64 bal = __builtin_amdgcn_ballot_w64(len > 0);
            if (len > 0) {
                // sources are marked at their LAST product: the source of product p is then simply
                // the number of marks before p (mbcnt), no own-bit correction
                const int sidx = (int)__builtin_amdgcn_mbcnt_hi((u32)(bal >> 32), __builtin_amdgcn_mbcnt_lo((u32)bal, (u32)nsrc));
                const int last = excl + len - 1;
                delta[sidx] = bs - excl;
                atomicOr(&starts[last >> 6], 1ull << (last & 63));
            }
            F += wave_bcast(inc, 63);
            nsrc += __popcll(bal);
        }
        wave_lds_fence();
        u64 sw = 0ull;
        if (lane < CHUNKS) { sw = starts[lane]; starts[lane] = 0ull; }
        const int sinc = wave_incl_scan(__popcll(sw));
        const int sbefore = sinc - __popcll(sw);
        int gaddr[CHUNKS];
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            const int p = c * 64 + lane;
            const u64 M = wave_bcast64(sw, c);
            const int before = wave_bcast(sbefore, c);
            int s = (int)__builtin_amdgcn_mbcnt_hi((u32)(M >> 32), __builtin_amdgcn_mbcnt_lo((u32)M, (u32)before));
            s = p < F ? s : 0;
            gaddr[c] = delta[s];
        }
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            const int p = c * 64 + lane;
            coln[c] = (u32)Bcol[gaddr[c] + (p < F ? p : 0)];
        }
        Fn = F;
        wave_lds_fence();   // delta is dead from here on (its reads have been consumed by the loads above)
    };
    load_acol(0);
    prefetch_extents();                                            // row 0
    load_acol(1);
    issue_gather(0);
    prefetch_extents();                                            // row 1
    load_acol(2);

    int my_cnt = 0;
    for (int kk = 0; kk < nmine; kk++) {
        u32 col[CHUNKS];
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) col[c] = coln[c];
        const int F = Fn;
        if (kk + 1 < nmine) {                                      // uniform
            issue_gather(kk + 1);
            prefetch_extents();                                    // row kk + 2
            load_acol(kk + 3);
        }

        // ---- hash set, first probe of every product: straight-line, all chunks in flight -----
        u32 *slot[CHUNKS];
        u32 old[CHUNKS];
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) slot[c] = tab + hash_slot<LOGH, WIDE>(col[c]);
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            old[c] = col[c];                                       // tail lanes: neither new nor unsettled
            if (c * 64 + lane < F) old[c] = atomicCAS(slot[c], kEmptySlot, col[c]);   // tail lanes: no LDS traffic
        }
        // A product is settled when its slot was empty (a NEW column, now stored: counted) or holds
        // its column (a duplicate: what `if (!xb[k])` skips at :38).  The others -- the slot holds
        // another column -- are squeezed together in LDS and probe on, one instruction per round for
        // all of them instead of one per chunk.
        int fresh = 0;                                             // wave-uniform: new columns stored so far
        int nuns = 0;
#pragma unroll
        for (int c = 0; c < CHUNKS; c++) {
            const u64 fb = __builtin_amdgcn_ballot_w64(old[c] == kEmptySlot);
            const u64 ub = __builtin_amdgcn_ballot_w64(old[c] != col[c]) & ~fb;
            fresh += __popcll(fb);
            if ((ub >> lane) & 1ull)
                unsettled[__builtin_amdgcn_mbcnt_hi((u32)(ub >> 32), __builtin_amdgcn_mbcnt_lo((u32)ub, (u32)nuns))] = col[c];
            nuns += __popcll(ub);
        }
        wave_lds_fence();
        u32 *slot0 = nullptr;                                      // last slot touched by this lane's key of batch 0
        for (int b0 = 0; b0 < nuns; b0 += 64) {                    // one trip unless > 64 keys are unsettled
            const bool have = b0 + lane < nuns;
            const u32 key = have ? unsettled[b0 + lane] : 0u;
            u32 hs = hash_slot<LOGH, WIDE>(key);
            bool go = have;
            while (__ballot(go)) {                                 // linear probing, ONE ds_cmpst per round
                u32 o = key;
                if (go) {
                    hs = (hs + 1u) & (u32)(H - 1);
                    o = atomicCAS(&tab[hs], kEmptySlot, key);
                }
                fresh += __popcll(__builtin_amdgcn_ballot_w64(o == kEmptySlot));
                go = go && o != kEmptySlot && o != key;
            }
            if (b0 == 0) slot0 = have ? tab + hs : nullptr;
            else if (have) unsettled[b0 + lane] = hs;              // remembered for the wipe (the key is done)
        }
        my_cnt = (lane == kk) ? fresh : my_cnt;                    // |C_i| = new columns; lane kk keeps row kk's
        // ---- wipe every slot this row touched (the reference's sparse reset, :48-50).  Touched but
        // not filled by this lane (a duplicate's, or another column's slot) is wiped by its owner too:
        // writing EMPTY twice is harmless now that nothing of this row probes any more.
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < CHUNKS; c++)
            if (c * 64 + lane < F) *slot[c] = kEmptySlot;
        if (slot0) *slot0 = kEmptySlot;
        for (int b0 = 64; b0 < nuns; b0 += 64)
            if (b0 + lane < nuns) tab[unsettled[b0 + lane]] = kEmptySlot;
        wave_lds_fence();
    }
    if (lane < nmine) cnt[r_row - row_begin] = my_cnt;
}

template <int CHUNKS>
static void launch_count_cfg(const int *Acol, const int *Brow, const int *Bcol, int cols, int2 *ab,
                             const RowRec *rec, int nrows, int row_begin, int *cnt, hipStream_t s)
{
    using Cfg = CountCfg<CHUNKS>;
    constexpr int kSpreadWaves = 256 * 8;
    int rpw = (int)((nrows + kSpreadWaves - 1) / kSpreadWaves);
    if (rpw > Cfg::RPW) rpw = Cfg::RPW;
    if (rpw < 1) rpw = 1;
    const long long rows_per_wg = (long long)Cfg::WAVES * rpw;
    const int grid = (int)((nrows + rows_per_wg - 1) / rows_per_wg);
    if (cols <= (1 << 24))
        hipLaunchKernelGGL((k_wave_count<CHUNKS, false>), dim3(grid), dim3(64 * Cfg::WAVES), 0, s,
                           Acol, Brow, Bcol, ab, rec, nrows, rpw, row_begin, cnt);
    else
        hipLaunchKernelGGL((k_wave_count<CHUNKS, true>), dim3(grid), dim3(64 * Cfg::WAVES), 0, s,
                           Acol, Brow, Bcol, ab, rec, nrows, rpw, row_begin, cnt);
}

void launch_wave_count(int bin, const int *Acol, const int *Brow, const int *Bcol, int cols, int2 *ab,
                       const RowRec *rec, int nrows, int row_begin, int *cnt, hipStream_t s)
{
    if (nrows <= 0) return;
    switch (bin) {
#define BSP_CASE(b) case b: launch_count_cfg<kWaveChunks[b]>(Acol, Brow, Bcol, cols, ab, rec, nrows, row_begin, cnt, s); break;
    BSP_CASE(1) BSP_CASE(2) BSP_CASE(3) BSP_CASE(4) BSP_CASE(5) BSP_CASE(6) BSP_CASE(7) BSP_CASE(8)
    BSP_CASE(9) BSP_CASE(10) BSP_CASE(11) BSP_CASE(12) BSP_CASE(13) BSP_CASE(14) BSP_CASE(15) BSP_CASE(16)
#undef BSP_CASE
    static_assert(kWaveBins == 16, "one case per capacity class");
    default: break;
    }
}

// ---------------------------------------------------------------------------------------
// Heavy rows keep the upper-bound placement: k_dense_rows accumulates AND reads out in the symbolic
// phase (its window bitmap is the expensive part; counting alone would cost almost the same), into
// a workspace sized by sum(min(F_i, cols)) over the heavy rows only, and this kernel moves each
// heavy row to its final place once C.row_ptr exists.  One workgroup per heavy row.
__global__ __launch_bounds__(256) void k_place_heavy(const int *__restrict__ tmp, const RowRec *__restrict__ rec,
                                                     const long long *__restrict__ recpre,
                                                     const long long *__restrict__ row_ptr, int row_begin,
                                                     int *__restrict__ col_idx)
{
    const RowRec q = rec[blockIdx.x];
    const int i = q.row - row_begin;
    const long long d0 = row_ptr[i];
    const int n = (int)(row_ptr[i + 1] - d0);
    const int *src = tmp + recpre[blockIdx.x];
    int *dst = col_idx + d0;
    for (int t = threadIdx.x; t < n; t += 256) dst[t] = src[t];
}

void launch_place_heavy(const int *tmp, const RowRec *rec, const long long *recpre, int nrows,
                        const long long *row_ptr, int row_begin, int *col_idx, hipStream_t s)
{
    if (nrows <= 0) return;
    hipLaunchKernelGGL(k_place_heavy, dim3(nrows), dim3(256), 0, s, tmp, rec, recpre, row_ptr, row_begin, col_idx);
}

}  // namespace bsp
