// dense_rows.hip -- heavy rows (F_i > 2048 products), one workgroup per A-row, three kernels over ONE gather:
//
//   k_dense_rows   a dense column bitmap in LDS, swept over column windows: the literal GPU form of the reference's accumulator
//                  (final/SpGEMM_mpi_omp.c:21,38-42) -- xb[k] becomes bit k of the LDS bitmap, test-and-set becomes ds_or_b32, and
//                  the quickSort of the row (:47) disappears because the bitmap is read out in column order.  Two shapes: 1024
//                  threads over windows of up to 2^20 columns (128 KiB), 512 threads over 2^18 (32 KiB).  When the columns exceed the
//                  window the row's products are gathered once per window;
//   k_rank_rows    (round 4) rows of at most 6144 products where the small shape would need two to four windows: a two-level rank
//                  bitmap sized by the ROW -- two sweeps and one read-out whatever the column count (see the kernel);
//   gather_sweep   the row's A-nonzeros one per thread; their B-row extents (left by the prepass) are scanned into QUAD offsets -- a
//                  quad is four consecutive entries of one B row, one 16-byte load -- and the quads are spread evenly over the
//                  threads in tiles: thread t takes quads t, t+T, ... and finds each one's source row by rank in a per-tile "starts"
//                  bitmap (the wave kernels' gather plan at workgroup scope), whatever the B-row lengths are.
// Also holds the exact flow's move of the heavy rows (k_place_heavy) and the compaction kernel that squeezes the
// upper-bound-placed rows into C.col_idx.
#include "kernels.hpp"
#include "wave.hpp"
#include <stdlib.h>
#include <type_traits>

namespace bsp {

// Two shapes of the same kernel (class kDenseBin / kMidBin, kernels.hpp):
//   1024 threads, window up to 2^20 columns (128 KiB): one workgroup per CU -- worth it for hub rows,
//        whose tens of thousands of products amortise the latency of every phase;
//    512 threads, window up to 2^18 columns (32 KiB): four workgroups (32 waves) per CU -- for the many
//        rows with thousands of products, which one workgroup per CU serialises phase by phase
//        (256 threads: the same four rows in flight with half the waves to cover their latencies).
constexpr int kDenseThreadsBig = 1024;
constexpr int kDenseThreadsMid = 512;
constexpr int kMidMinWaves = 8;              // four 8-wave workgroups per CU: 64 VGPRs
constexpr int kBigMinWaves = 4;
constexpr int kDenseMaxWords = 16384;        // 64-bit words per window = 2^20 columns = 128 KiB
constexpr int kMidMaxWords = 4096;           // 4096 words = 2^18 columns = 32 KiB
constexpr int kDenseWordBits = 12;           // 64-column words with at least this many outputs are emitted by a whole wave
constexpr int kDenseQuadsPerThreadBig = 16;  // quads (16-byte pieces of a B row) per thread and tile: 64 / 32 products
constexpr int kDenseQuadsPerThreadMid = 8;   //   (the small shape's four workgroups share the CU's LDS: a smaller plan)
constexpr int kDenseInFlightBig = 8;         // 16-byte B.col_idx loads a thread keeps in flight
constexpr int kDenseInFlightMid = 4;         //   (64 VGPRs)

struct __attribute__((packed, aligned(4))) Int4U { int x, y, z, w; };   // 16 B, only dword aligned
struct __attribute__((packed, aligned(4))) Int2U { int x, y; };

// ---------------------------------------------------------------------------------------
// The heavy rows' GATHER, shared by the windowed kernel and the rank kernel below: one sweep over all the products
// of the row, `ins(quad, valid lanes)` called once per quad.
//
// One source (A-nonzero) per thread, kThreads at a time (a "batch").  The unit is the QUAD: four consecutive entries
// of one B row, one 16-byte load.  A block scan of the sources' quad counts gives every source its place in the batch's
// quad order; the non-empty sources are squeezed into a list of (B address - 4 * first quad, B end).  The quads are
// taken in TILES of kThreads * kQPT: a "starts" bitmap over the tile marks where each source begins, one wave turns its
// words into running source counts, and quad t finds its source by rank -- word, count, popcount: two independent LDS
// reads and a dependent one, the same for B rows of 3 and of 30000 entries -- so that every thread keeps kInFlight
// 16-byte loads in the air.  (Rounds 1-3 looked up every PRODUCT this way, three LDS reads and one 4-byte load each;
// the heavy classes were bound by exactly those, not by memory: profiles/r04_heavy_ablation.log.)
// The last quad of a source is the four entries that END at the row's end: it overlaps the quad before it (the
// accumulators are sets: inserting a column twice is harmless) and for a source of one to three entries it begins
// before the source -- those lanes are masked.  No load ever passes the end of B.col_idx.
// A row of one batch and one tile KEEPS its plan (sd, tw, tpre in LDS) for the later sweeps: they then start at the
// loads -- no extents, no block scan, no tile bitmap, none of their barriers (each of these phases is a latency the
// row's few waves cannot hide).
struct GatherState {
    long long QB = 0;            // quads of the batch
    bool plan_kept = false;      // uniform
    int buf = 0;                 // which of tw / tpre the current tile reads
};

template <int kThreads, int kQPT>
struct GatherLds {
    static constexpr int kWaves = kThreads / 64;
    static constexpr int kTileQ = kThreads * kQPT;                 // quads per tile
    static constexpr int kTileWords = kTileQ / 32;
    static_assert(kTileWords % 64 == 0 && kTileWords <= kThreads, "one wave scans the tile's words, blocked");
    int wcnt[kWaves];
    long long wsum[kWaves];
    int2 sd[kThreads];            // non-empty sources of the batch: (B address - 4 * first quad, B end address)
    u32 tb[kTileWords];           // starts of the sources inside the tile being planned (all zero between tiles)
    u32 tw[2][kTileWords];        // ... as the gather reads them: two tiles, so that the next plan never waits for the slowest gather
    int tpre[2][kTileWords];      // (sources begun before word w) - 1
};

// before the kernel's first barrier
template <int kThreads, int kQPT>
__device__ __forceinline__ void gather_init(GatherLds<kThreads, kQPT> &L)
{
    if ((int)threadIdx.x < GatherLds<kThreads, kQPT>::kTileWords) L.tb[threadIdx.x] = 0u;
}

template <int kThreads, int kQPT, int kInFlight, typename Ins>
__device__ __forceinline__ void gather_sweep(GatherLds<kThreads, kQPT> &L, GatherState &g, const int2 *__restrict__ ab,
                                             const int *__restrict__ Bcol, int nnzB, int a0, int a1, bool first_sweep, Ins ins)
{
    using LT = GatherLds<kThreads, kQPT>;
    constexpr int kWaves = LT::kWaves, kTileQ = LT::kTileQ, kTileWords = LT::kTileWords;
    static_assert(kQPT % kInFlight == 0 && kInFlight % 4 == 0, "whole steps; the last step in quarters");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (a row of several batches -- a hub row has thousands of sources -- loads the extents of batch j+1 while batch j is
    // gathered: one trip to memory less per batch on the row's critical path, and a hub row has the CU to itself)
    int2 e_next = make_int2(0, 0);
    if (!g.plan_kept && a0 + tid < a1) e_next = ab[a0 + tid];
    for (int ja = a0; ja < a1; ja += kThreads) {
        int2 e = make_int2(0, 0);
        int nq = 0, cidx = 0;
        long long qexcl = 0;
        if (!g.plan_kept) {
            e = e_next;
            e_next = make_int2(0, 0);
            if (ja + kThreads + tid < a1) e_next = ab[ja + kThreads + tid];
            nq = (int)(((u32)e.y + 3u) >> 2);
            const int inc = wave_incl_scan(nq);
            const u64 nonempty = __ballot(nq > 0);
            if (lane == 63) L.wsum[wave] = (long long)inc;
            if (lane == 0) L.wcnt[wave] = __popcll(nonempty);
            __syncthreads();
            qexcl = (long long)(inc - nq);
            g.QB = 0;
            cidx = __popcll(nonempty & mask_lt(lane));
            for (int k = 0; k < kWaves; k++) {
                const long long t = L.wsum[k];
                const int c = L.wcnt[k];
                if (k < wave) { qexcl += t; cidx += c; }
                g.QB += t;
            }
            if (nq > 0) L.sd[cidx] = make_int2((int)((u32)e.x - 4u * (u32)qexcl), e.x + e.y);   // (mod 2^32: the sum is a B address again)
            if (g.QB == 0) __syncthreads();                        // (no tile: nothing else orders this batch's wsum reads before the next batch's writes)
        }
        int carry = -1;                                            // wave 0: (sources begun before the tile) - 1
        for (long long T0 = 0; T0 < g.QB; T0 += kTileQ) {
            if (!g.plan_kept) {
                g.buf ^= 1;
                if (nq > 0 && qexcl >= T0 && qexcl < T0 + kTileQ) {    // the source begins in this tile
                    const int rel = (int)(qexcl - T0);
                    atomicOr(&L.tb[rel >> 5], 1u << (rel & 31));
                }
                __syncthreads();
                if (wave == 0) {
                    constexpr int WPL = kTileWords / 64;               // words per lane, blocked
                    u32 x[WPL];
                    int c[WPL], run = 0;
#pragma unroll
                    for (int k = 0; k < WPL; k++) {
                        x[k] = L.tb[lane * WPL + k];
                        L.tb[lane * WPL + k] = 0u;
                        c[k] = run;
                        run += __popc(x[k]);
                    }
                    const int wi = wave_incl_scan(run);
#pragma unroll
                    for (int k = 0; k < WPL; k++) {
                        L.tw[g.buf][lane * WPL + k] = x[k];
                        L.tpre[g.buf][lane * WPL + k] = carry + wi - run + c[k];
                    }
                    carry += wave_bcast(wi, 63);
                }
                __syncthreads();
            }
            const int nqt = (g.QB - T0 < kTileQ) ? (int)(g.QB - T0) : kTileQ;
            const u32 *twb = L.tw[g.buf];
            const int *tpb = L.tpre[g.buf];
            const u32 T0lo = 4u * (u32)T0;
            // a step takes kInFlight quads per thread; the LAST step of a tile is specialised for the number of slots that still
            // hold quads for anybody (workgroup-uniform): it is half empty on average, and a row of 3 K products fills 750 of a
            // step's 2048 quads -- the empty slots used to cost their look-ups and inserts all the same
            auto step = [&](int k0, auto nu_c) {
                constexpr int NU = decltype(nu_c)::value;
                int base[NU];
                u32 vmask[NU];                                     // lanes of the quad that are entries of this source not yet taken
#pragma unroll
                for (int u = 0; u < NU; u++) {
                    const int t = k0 + u * kThreads + tid;
                    const bool ok = t < nqt;
                    const int tt = ok ? t : 0;
                    const u32 w = twb[tt >> 5];
                    const int src = tpb[tt >> 5] + __popc(w & ((2u << (tt & 31)) - 1u));
                    const int2 sq = L.sd[src < 0 ? 0 : src];
                    const int qs = (int)((u32)sq.x + T0lo + 4u * (u32)tt);   // first entry of the quad
                    int b = qs < sq.y - 4 ? qs : sq.y - 4;
                    b = b < 0 ? 0 : b;
                    base[u] = b;
                    u32 m = 0u;
#pragma unroll
                    for (int k = 0; k < 4; k++) m |= (ok && b + k >= qs && b + k < sq.y) ? (1u << k) : 0u;
                    vmask[u] = m;
                }
                Int4U cv[NU];
#pragma unroll
                for (int u = 0; u < NU; u++) {
                    if (nnzB >= 4) {                               // (uniform)
                        cv[u] = *reinterpret_cast<const Int4U *>(Bcol + base[u]);      // only dword aligned
                    } else {                                       // a B of one to three entries: base is 0, no vector load fits
                        cv[u].x = Bcol[0];
                        cv[u].y = nnzB > 1 ? Bcol[1] : 0;
                        cv[u].z = nnzB > 2 ? Bcol[2] : 0;
                        cv[u].w = 0;
                    }
                }
#pragma unroll
                for (int u = 0; u < NU; u++) ins(cv[u], vmask[u], u);
            };
            for (int k0 = 0; k0 < nqt; k0 += kInFlight * kThreads) {
                const int left = nqt - k0;                         // (uniform)
                if (left > (3 * kInFlight / 4) * kThreads) step(k0, std::integral_constant<int, kInFlight>());
                else if (left > (kInFlight / 2) * kThreads) step(k0, std::integral_constant<int, 3 * kInFlight / 4>());
                else if (left > (kInFlight / 4) * kThreads) step(k0, std::integral_constant<int, kInFlight / 2>());
                else step(k0, std::integral_constant<int, kInFlight / 4>());
            }
        }
        __syncthreads();
    }
    if (first_sweep) g.plan_kept = (a1 - a0 <= kThreads) && g.QB <= kTileQ && g.QB > 0;
}

// One quad into a bitmap: entry k goes to bit b_k of word w_k when i_k.  The quad's columns ascend, so the entries of one
// word are neighbours, and the first of each run ORs the whole run -- one LDS atomic per word touched instead of one per
// product (the dense heads of hub B rows put up to 32 lanes' products into ONE word: same-address atomics serialise).
// Correct for any order (an unsorted B row only merges less).
// `windowed`: the entries are filtered by a column window, and a wave whose 64 quads all miss it leaves at once.
__device__ __forceinline__ void insert_quad(u32 *tgt, bool i0, bool i1, bool i2, bool i3, u32 w0, u32 w1, u32 w2, u32 w3,
                                            u32 b0, u32 b1, u32 b2, u32 b3, bool windowed = false)
{
    if (windowed && !__ballot(i0 | i1 | i2 | i3)) return;          // (wave-uniform) nothing of these 64 quads falls into the window
    w0 = i0 ? w0 : 0xfffffff0u, w1 = i1 ? w1 : 0xfffffff1u, w2 = i2 ? w2 : 0xfffffff2u, w3 = i3 ? w3 : 0xfffffff3u;
    const u32 m2 = b2 | (w3 == w2 ? b3 : 0u);
    const u32 m1 = b1 | (w2 == w1 ? m2 : 0u);
    const u32 m0 = b0 | (w1 == w0 ? m1 : 0u);
    if (i0) atomicOr(&tgt[w0], m0);
    if (i1 && w1 != w0) atomicOr(&tgt[w1], m1);
    if (i2 && w2 != w1) atomicOr(&tgt[w2], m2);
    if (i3 && w3 != w2) atomicOr(&tgt[w3], b3);
}

// MASKED: C = F .* (A*B) (SpGEMM_masked, final/SpGEMM_mpi_omp.c:232-288).  The reference presets
// its flag array so that only columns of F's row can be appended (:253-255); here the window
// holds two bitmaps, P (products) and K (kept): after the gather every column of F's row that is
// set in P is set in K, and K is what gets read out.
template <bool MASKED, int kDenseThreads>
__global__ __launch_bounds__(kDenseThreads, (kDenseThreads == kDenseThreadsBig ? kBigMinWaves : kMidMinWaves)) void k_dense_rows(const int2 *__restrict__ ab,
                                                              const int *__restrict__ Bcol, int nnzB,
                                                              int cols, int wwords,
                                                              const RowRec *__restrict__ rec,
                                                              const long long *__restrict__ recpre,
                                                              int row_begin,
                                                              int *__restrict__ tmp,
                                                              int *__restrict__ cnt,
                                                              const int *__restrict__ Frow,
                                                              const int *__restrict__ Fcol)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    u64 *bmP = reinterpret_cast<u64 *>(lds_raw);                       // products
    u32 *bm32 = reinterpret_cast<u32 *>(lds_raw);
    u64 *bm = MASKED ? bmP + wwords : bmP;                             // what is read out (K or P)
    u32 *bmK32 = reinterpret_cast<u32 *>(bm);
    constexpr int kWaves = kDenseThreads / 64;
    constexpr int kQPT = kDenseThreads == kDenseThreadsBig ? kDenseQuadsPerThreadBig : kDenseQuadsPerThreadMid;
    constexpr int kInFlight = kDenseThreads == kDenseThreadsBig ? kDenseInFlightBig : kDenseInFlightMid;   // 16-byte loads a thread keeps in flight
    __shared__ GatherLds<kDenseThreads, kQPT> G;
    __shared__ int wtot[kWaves];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int t = tid; t < (MASKED ? 2 * wwords : wwords); t += kDenseThreads) bmP[t] = 0ull;
    gather_init(G);
    __syncthreads();

    const RowRec q = rec[blockIdx.x];
    const int i = q.row;
    const int a0 = q.a0, a1 = q.a0 + q.alen;
    int *out = tmp + recpre[blockIdx.x];
    const long long W = (long long)wwords * 64;
    const int nwin = (int)(((long long)cols + W - 1) / W);
    int total = 0;

    GatherState g;
    for (int win = 0; win < nwin; win++) {
        const long long lo = (long long)win * W;
        const int lo32 = (int)lo;
        gather_sweep<kDenseThreads, kQPT, kInFlight>(G, g, ab, Bcol, nnzB, a0, a1, win == 0, [&](const Int4U &v, u32 vm, int) {
            const u32 c0 = (u32)(v.x - lo32), c1 = (u32)(v.y - lo32), c2 = (u32)(v.z - lo32), c3 = (u32)(v.w - lo32);
            insert_quad(bm32, (vm & 1u) && c0 < (u32)W, (vm & 2u) && c1 < (u32)W, (vm & 4u) && c2 < (u32)W, (vm & 8u) && c3 < (u32)W,   // (columns below the window wrap to huge values)
                        c0 >> 5, c1 >> 5, c2 >> 5, c3 >> 5, 1u << (c0 & 31), 1u << (c1 & 31), 1u << (c2 & 31), 1u << (c3 & 31), nwin > 1);
        });
        if (MASKED) {
            // keep the product bits that F's row admits, then wipe P for the next window / row
            const int f0 = Frow[i], f1 = Frow[i + 1];
            for (int k = f0 + tid; k < f1; k += kDenseThreads) {
                const long long c = (long long)Fcol[k] - lo;
                if (c >= 0 && c < W && ((bm32[c >> 5] >> (c & 31)) & 1u)) atomicOr(&bmK32[c >> 5], 1u << (c & 31));
            }
            __syncthreads();
            for (int t = tid; t < wwords; t += kDenseThreads) bmP[t] = 0ull;
            __syncthreads();
        }
        // read-out in column order: wave w owns the words [w*wpw, (w+1)*wpw); a step takes 64*kWpl
        // consecutive words, lane l the kWpl words behind 64-bit word kWpl*l of the step -- a lane's
        // outputs are one contiguous piece of the row, the step's pieces follow each other.  kWpl = 4:
        // ONE wave scan per 256 words (it was one per 64: in a window that is mostly empty -- a row with a
        // few thousand products over 2^18 columns -- the scans were two thirds of the kernel's VALU work).
        constexpr int kWpl = 4;
        constexpr int kStepWords = 64 * kWpl;
        constexpr int kWavesPerWg = kDenseThreads / 64;
        const int wpw = ((wwords + kWavesPerWg - 1) / kWavesPerWg + kStepWords - 1) / kStepWords * kStepWords;
        const int wbeg = wave * wpw;
        const int wend = (wbeg + wpw < wwords) ? wbeg + wpw : wwords;
        int c = 0;
        for (int w = wbeg + lane; w < wend; w += 64) c += __popcll(bm[w]);
        const int inc = wave_incl_scan(c);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int off = 0, btotal = 0;
        for (int k = 0; k < kDenseThreads / 64; k++) {
            const int t = wtot[k];
            if (k < wave) off += t;
            btotal += t;
        }
        int run = total + off;                                 // wave-uniform output cursor
        for (int w0 = wbeg; w0 < wend; w0 += kStepWords) {
            const int wl = w0 + kWpl * lane;                   // this lane's first word
            u64 m[kWpl];
            int cw = 0;
#pragma unroll
            for (int k = 0; k < kWpl; k++) {
                m[k] = 0ull;
                if (wl + k < wend) { m[k] = bm[wl + k]; bm[wl + k] = 0ull; }
                cw += __popcll(m[k]);
            }
            const int iw = wave_incl_scan(cw);
            const int step_total = wave_bcast(iw, 63);
            if (step_total == 0) continue;                     // uniform: an empty stretch of the window
            const int base = (int)(lo + (long long)wl * 64);
            // A step of few outputs is STAGED: the lanes expand their words into the step's own 2 KiB of the window (read and
            // cleared just above, by this wave) and the wave streams the piece out coalesced.  Written straight from the
            // per-lane loop, every store instruction of such a step touches up to 64 different 64-byte sectors -- those
            // stores were 25-30 % of the small shape's time (profiles/r04_heavy_ablation.log, part 6).
            const int stage_cap = 2 * ((wend - w0 < kStepWords) ? wend - w0 : kStepWords);    // 32-bit entries
            if (step_total <= stage_cap) {                     // (uniform)
                u32 *stage = reinterpret_cast<u32 *>(bm + w0);
                int p = iw - cw;
#pragma unroll
                for (int k = 0; k < kWpl; k++) {
                    u64 mk = m[k];
                    while (mk) {
                        stage[p++] = (u32)((base + 64 * k) | (int)__builtin_ctzll(mk));
                        mk &= mk - 1ull;
                    }
                }
                wave_lds_fence();
                for (int j = lane; j < step_total; j += 64) {
                    const u32 v = stage[j];
                    stage[j] = 0u;                             // the window is all zero again
                    out[run + j] = (int)v;
                }
                wave_lds_fence();
                run += step_total;
                continue;
            }
            int pos = run + iw - cw;
#pragma unroll
            for (int k = 0; k < kWpl; k++) {
                const int ck = __popcll(m[k]);
                // dense words (hub columns: up to 64 bits set) are written by the whole wave, one word
                // per store instruction, lane b holding bit b; the per-lane loop below then never runs
                // longer than kDenseWordBits trips while the other lanes idle
                u64 crowded = __ballot(ck >= kDenseWordBits);
                while (crowded) {
                    const int src = (int)__builtin_ctzll(crowded);
                    crowded &= crowded - 1ull;
                    const u64 mw = wave_bcast64(m[k], src);
                    const int pw = wave_bcast(pos, src);
                    const int bw = wave_bcast(base, src) + 64 * k;
                    if ((mw >> lane) & 1ull) out[pw + __popcll(mw & mask_lt(lane))] = bw | lane;
                }
                u64 mk = (ck >= kDenseWordBits) ? 0ull : m[k];
                int p = pos;
                while (mk) {                                   // two outputs per store instruction (8 bytes, only dword aligned)
                    const int v0 = (base + 64 * k) | (int)__builtin_ctzll(mk);
                    mk &= mk - 1ull;
                    if (mk) {
                        Int2U v2;
                        v2.x = v0;
                        v2.y = (base + 64 * k) | (int)__builtin_ctzll(mk);
                        mk &= mk - 1ull;
                        *reinterpret_cast<Int2U *>(out + p) = v2;
                        p += 2;
                    } else {
                        out[p++] = v0;
                    }
                }
                pos += ck;
            }
            run += step_total;
        }
        total += btotal;
        __syncthreads();
    }
    if (tid == 0) cnt[i - row_begin] = total;
}

// ---------------------------------------------------------------------------------------
// RANK ROWS (class kRankBin): rows of 2048 < F_i <= kRankCap products when the column range is several windows of the
// small dense shape.  The workgroup's LDS holds a two-level RANK bitmap instead of a dense one -- `top`, one bit per
// 32-column word of the whole column range, and one 32-bit slot per SET top bit, addressed by the bit's rank (the
// one-wave kernels' accumulator at workgroup scope, wave_rows.inc).  Slots are as many as the row has distinct words
// (<= F_i), not as many as the matrix has columns: the row is done in TWO sweeps over its products (top bits; slot bits)
// and ONE read-out proportional to the row, whatever the column count is -- the windowed shape runs one sweep and one
// 4096-word read-out per 2^18 columns, four of each at 2^20 columns for a row that fills 1-2 % of every window, and
// every one of them is a trip to memory that the row's eight waves wait for (profiles/r04_heavy_ablation.log).
// LDS per workgroup: top (cols / 32 bits) + ranks (32 bits per top word) + kRankCap slots: 32 KiB at 2^20 columns,
// beside the gather plan -- four workgroups per CU, as many as the windowed shape.
constexpr int kRankThreads = 512;
constexpr int kRankSlotsPerThread = kRankCap / kRankThreads;
static_assert(kRankCap % kRankThreads == 0, "whole slots per thread");
constexpr int kRankQPT = 8, kRankInFlight = 4;
constexpr int kRankSpan = 1 << 20;           // columns one pass covers: the top bitmap's reach (4 KiB of top bits)

template <bool kSpans>   // false: the column range is one span (the common case: one pass, its quads kept in registers)
__global__ __launch_bounds__(kRankThreads, 8) void k_rank_rows(const int2 *__restrict__ ab, const int *__restrict__ Bcol, int nnzB,
                                                               int cols, int topw,
                                                               const RowRec *__restrict__ rec,
                                                               const long long *__restrict__ recpre,
                                                               int row_begin, int *__restrict__ tmp, int *__restrict__ cnt)
{
    // (the class is bound by LDS instruction issue -- profiles/r04_rank_rows_phases.log -- so the layout is chosen for few LDS
    // instructions: a top word and its rank are one 8-byte pair, one read in sweep 2)
    extern __shared__ __attribute__((aligned(16))) unsigned char lds_raw[];
    uint2 *tp = reinterpret_cast<uint2 *>(lds_raw);                             // [topw] x: bit (c >> 5) of the span, 32 per word; y: set bits before the word
    u32 *tp32 = reinterpret_cast<u32 *>(lds_raw);
    u32 *S = tp32 + 2 * topw;                                                   // [kRankCap] slots; later the staged row
    constexpr int kWaves = kRankThreads / 64;
    constexpr int SPT = kRankSlotsPerThread;
    __shared__ GatherLds<kRankThreads, kRankQPT> G;
    __shared__ int wtot[kWaves];
    __shared__ unsigned short fw[kRankThreads];                                 // top word that holds slot t * SPT
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nclear = topw + kRankCap / 2;                                     // 8-byte words of the accumulator
    {
        u64 *z = reinterpret_cast<u64 *>(lds_raw);
        for (int t = tid; t < nclear; t += kRankThreads) z[t] = 0ull;
    }
    gather_init(G);
    __syncthreads();

    const RowRec q = rec[blockIdx.x];
    const int a0 = q.a0, a1 = q.a0 + q.alen;
    int *out = tmp + recpre[blockIdx.x];

    GatherState g;
    // a row whose quads are one step of the gather keeps them in registers for every later sweep: no plan look-ups, no loads
    Int4U hq[kRankInFlight];
    u32 hm[kRankInFlight];
#pragma unroll
    for (int u = 0; u < kRankInFlight; u++) {                      // (slots the gather's last step leaves out stay empty)
        hq[u].x = hq[u].y = hq[u].z = hq[u].w = 0;
        hm[u] = 0u;
    }
    bool held = false;                                             // uniform
    // The column range is taken in SPANS of 2^20 columns (the top bitmap's reach): one for the matrices the class was built
    // for, up to sixteen on wider ones -- where the small dense shape would sweep and read out 4 * sixteen windows.
    const int nspans = kSpans ? (int)(((long long)cols + kRankSpan - 1) / kRankSpan) : 1;
    int total = 0;
    for (int sp = 0; sp < nspans; sp++) {
        const u32 lo = kSpans ? (u32)sp * (u32)kRankSpan : 0u;
        // ---- sweep 1: the top bits -----------------------------------------------------------------------------------
        auto top_bits = [&](const Int4U &v, u32 vm, int u) {
            if (!kSpans) {
                hq[u] = v;
                hm[u] = vm;
            }
            const u32 c0 = (u32)v.x - lo, c1 = (u32)v.y - lo, c2 = (u32)v.z - lo, c3 = (u32)v.w - lo;   // (columns below the span wrap to huge values)
            insert_quad(tp32, (vm & 1u) && (!kSpans || c0 < (u32)kRankSpan), (vm & 2u) && (!kSpans || c1 < (u32)kRankSpan),
                        (vm & 4u) && (!kSpans || c2 < (u32)kRankSpan), (vm & 8u) && (!kSpans || c3 < (u32)kRankSpan), (c0 >> 10) * 2u, (c1 >> 10) * 2u, (c2 >> 10) * 2u, (c3 >> 10) * 2u,
                        1u << ((c0 >> 5) & 31), 1u << ((c1 >> 5) & 31), 1u << ((c2 >> 5) & 31), 1u << ((c3 >> 5) & 31), kSpans);
        };
        gather_sweep<kRankThreads, kRankQPT, kRankInFlight>(G, g, ab, Bcol, nnzB, a0, a1, sp == 0, top_bits);
        if (!kSpans) held = g.plan_kept && g.QB <= kRankInFlight * kRankThreads;   // (only ever used by sweep 2 of the single span)
        // ---- ranks of the top bits: thread t owns the words [t*WPT, (t+1)*WPT) -----------------------------------------
        int nslots = 0, spt = SPT;
        {
            const int WPT = topw / kRankThreads;                   // 1 or 2
            u32 x[2];
            int c[2], run = 0;
#pragma unroll
            for (int k = 0; k < 2; k++) {
                x[k] = k < WPT ? tp[tid * WPT + k].x : 0u;
                c[k] = run;
                run += __popc(x[k]);
            }
            const int inc = wave_incl_scan(run);
            if (lane == 63) wtot[wave] = inc;
            __syncthreads();
            int off = 0;
            for (int k = 0; k < kWaves; k++) {
                const int t = wtot[k];
                if (k < wave) off += t;
                nslots += t;
            }
            // slots per thread of the read-out: the row's slots spread evenly over the workgroup (a row of 2500 slots: five
            // per thread on all eight waves, not twelve on the first four)
            spt = (nslots + kRankThreads - 1) / kRankThreads;
            spt = spt < 1 ? 1 : (spt > SPT ? SPT : spt);
#pragma unroll
            for (int k = 0; k < 2; k++)
                if (k < WPT) {
                    const int pre = off + inc - run + c[k], end = pre + __popc(x[k]);
                    tp[tid * WPT + k].y = (u32)pre;
                    for (int j = (pre + spt - 1) / spt; j * spt < end && j < kRankThreads; j++) fw[j] = (unsigned short)(tid * WPT + k);   // (the first slot of thread j lies in this word)
                }
            __syncthreads();
        }
        // ---- sweep 2: bit (c & 31) of the slot whose index is the rank of top bit (c >> 5) ----------------------------
        auto slot_bits = [&](const Int4U &v, u32 vm, int) {
            const u32 c0 = (u32)v.x - lo, c1 = (u32)v.y - lo, c2 = (u32)v.z - lo, c3 = (u32)v.w - lo;
            const bool i0 = (vm & 1u) && (!kSpans || c0 < (u32)kRankSpan), i1 = (vm & 2u) && (!kSpans || c1 < (u32)kRankSpan);
            const bool i2 = (vm & 4u) && (!kSpans || c2 < (u32)kRankSpan), i3 = (vm & 8u) && (!kSpans || c3 < (u32)kRankSpan);
            const uint2 x0 = tp[i0 ? c0 >> 10 : 0u], x1 = tp[i1 ? c1 >> 10 : 0u], x2 = tp[i2 ? c2 >> 10 : 0u], x3 = tp[i3 ? c3 >> 10 : 0u];
            const u32 r0 = x0.y + __popc(__builtin_amdgcn_ubfe(x0.x, 0u, (c0 >> 5) & 31)), r1 = x1.y + __popc(__builtin_amdgcn_ubfe(x1.x, 0u, (c1 >> 5) & 31));
            const u32 r2 = x2.y + __popc(__builtin_amdgcn_ubfe(x2.x, 0u, (c2 >> 5) & 31)), r3 = x3.y + __popc(__builtin_amdgcn_ubfe(x3.x, 0u, (c3 >> 5) & 31));
            // (r < kRankCap always on consistent operands: slots <= F_i <= kRankCap; a rewritten operand is cut off, not LDS overrun)
            insert_quad(S, i0 && r0 < (u32)kRankCap, i1 && r1 < (u32)kRankCap, i2 && r2 < (u32)kRankCap, i3 && r3 < (u32)kRankCap, r0, r1, r2, r3,
                        1u << (c0 & 31), 1u << (c1 & 31), 1u << (c2 & 31), 1u << (c3 & 31), kSpans);
        };
        if (held) {
#pragma unroll
            for (int u = 0; u < kRankInFlight; u++)
                if ((long long)u * kRankThreads < g.QB) slot_bits(hq[u], hm[u], u);   // (uniform: the slots the gather's one step filled)
            __syncthreads();
        } else {
            gather_sweep<kRankThreads, kRankQPT, kRankInFlight>(G, g, ab, Bcol, nnzB, a0, a1, false, slot_bits);
        }
        // ---- read-out: slots are in column order.  Thread t owns the slots [t*SPT, (t+1)*SPT): their masks go to registers,
        // one block scan gives the thread its place in the row, the top word of its first slot was noted by the rank scan and
        // the others follow by walking the top bits; the columns are staged in LDS (over the slots, which every thread has
        // read by then) and streamed out coalesced.
        u32 m[SPT];
        int mine = 0;
#pragma unroll
        for (int k = 0; k < SPT; k++) {
            m[k] = k < spt ? S[tid * spt + k] : 0u;
            mine += __popc(m[k]);
        }
        const int inc = wave_incl_scan(mine);
        if (lane == 63) wtot[wave] = inc;
        __syncthreads();
        int pos = inc - mine, stotal = 0;
        for (int k = 0; k < kWaves; k++) {
            const int t = wtot[k];
            if (k < wave) pos += t;
            stotal += t;
        }
        if (nslots > kRankCap) nslots = kRankCap;
        const int s0 = tid * spt;
        if (s0 < nslots) {
            int t = fw[tid];
            const uint2 first = tp[t];
            u32 rem = first.x;
            for (int skip = s0 - (int)first.y; skip > 0; skip--) rem &= rem - 1u;
#pragma unroll
            for (int k = 0; k < SPT; k++) {
                if (k < spt && s0 + k < nslots) {
                    while (!rem && t + 1 < topw) rem = tp[++t].x;
                    const u32 base = lo + (((u32)t << 10) | ((u32)__builtin_ctz(rem | 0x80000000u) << 5));
                    rem &= rem - 1u;
                    u32 mk = m[k];
                    while (mk) {
                        if (pos < kRankCap) S[stage_swz(pos)] = base | (u32)__builtin_ctz(mk);   // (always, on consistent operands)
                        pos++;
                        mk &= mk - 1u;
                    }
                }
            }
        }
        __syncthreads();
        if (total + stotal > q.f) stotal = q.f > total ? q.f - total : 0;          // (never, on consistent operands: the row's room is F_i <= kRankCap)
        for (int t = tid; t < stotal; t += kRankThreads) __builtin_nontemporal_store((int)S[stage_swz(t)], out + total + t);
        total += stotal;
        if (sp + 1 < nspans) {                                     // the accumulator all zero again for the next span
            __syncthreads();
            u64 *z = reinterpret_cast<u64 *>(lds_raw);
            for (int t = tid; t < nclear; t += kRankThreads) z[t] = 0ull;
            __syncthreads();
        }
    }
    if (tid == 0) cnt[q.row - row_begin] = total;
}

template <bool MASKED, int THREADS>
static hipError_t launch_dense_impl(const int2 *ab, const int *Bcol, long long nnzB, int cols, const RowRec *rec,
                                    const long long *recpre, int nrows, int row_begin, int *tmp, int *cnt,
                                    const int *Frow, const int *Fcol, hipStream_t s)
{
    if (nrows <= 0) return hipSuccess;
    const long long cap_words = THREADS == kDenseThreadsBig ? kDenseMaxWords : kMidMaxWords;
    const long long max_words = MASKED ? cap_words / 2 : cap_words;   // two bitmaps share the window
    long long words = ((long long)cols + 63) / 64;
    if (words > max_words) words = max_words;
    if (words < 1) words = 1;
    const int bytes = (int)words * 8 * (MASKED ? 2 : 1);
    // the attribute belongs to the (kernel, device) pair: a process may hold contexts on several GPUs
    static bool attr_set[64] = {};
    int dev = 0;
    if (hipError_t e = hipGetDevice(&dev)) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_dense_rows<MASKED, THREADS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap_words * 8);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((k_dense_rows<MASKED, THREADS>), dim3(nrows), dim3(THREADS), bytes, s, ab, Bcol,
                       (int)(nnzB > 0x7fffffffll ? 0x7fffffffll : nnzB), cols, (int)words, rec, recpre, row_begin, tmp, cnt, Frow, Fcol);
    return hipGetLastError();
}

// Hub rows in order of decreasing products (longest processing time first): one workgroup per row, one per
// CU, dispatched in list order -- in row order the largest row (it alone is most of the class's critical
// path: 2 M products on one CU) may start last.  n <= kHeavySortMax: ranks by counting, each thread its own.
__global__ __launch_bounds__(256) void k_order_heavy(const RowRec *__restrict__ rec, const long long *__restrict__ recpre,
                                                     int n, RowRec *__restrict__ rec_out, long long *__restrict__ pre_out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const RowRec me = rec[i];
    int rank = 0;
    for (int j = 0; j < n; j++) {
        const int f = rec[j].f;
        rank += (f > me.f || (f == me.f && j < i)) ? 1 : 0;
    }
    rec_out[rank] = me;
    pre_out[rank] = recpre[i];
}
void launch_order_heavy(const RowRec *rec, const long long *recpre, int n, RowRec *rec_out, long long *pre_out, hipStream_t s)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_order_heavy, dim3((n + 255) / 256), dim3(256), 0, s, rec, recpre, n, rec_out, pre_out);
}

// rows of the rank class are told apart by the prepass (bin_of): rank_cap_for_cols(cols) products or fewer
int rank_cap_for_cols(long long cols)
{
    // (development switch, read once: BSPGEMM_RANK_ROWS=0 no rank class, =2 also where the small shape needs ONE window)
    static const int mode = [] { const char *e = getenv("BSPGEMM_RANK_ROWS"); return e ? atoi(e) : 1; }();
    if (mode <= 0 || cols > (1ll << 24)) return 0;                 // (one pass per 2^20 columns: sixteen at most; wider matrices keep the dense shapes)
    if (mode == 1 && cols <= (1ll << 18)) return 0;
    return kRankCap;
}

static hipError_t launch_rank_rows(const int2 *ab, const int *Bcol, long long nnzB, int cols, const RowRec *rec,
                                   const long long *recpre, int nrows, int row_begin, int *tmp, int *cnt, hipStream_t s)
{
    if (nrows <= 0) return hipSuccess;
    const long long span = cols < kRankSpan ? cols : kRankSpan;
    const int topw = (int)(((span + 1023) >> 10) + kRankThreads - 1) / kRankThreads * kRankThreads;   // whole words per thread
    const int bytes = topw * 8 + kRankCap * 4;
    if (cols > kRankSpan)
        hipLaunchKernelGGL(k_rank_rows<true>, dim3(nrows), dim3(kRankThreads), bytes, s, ab, Bcol,
                           (int)(nnzB > 0x7fffffffll ? 0x7fffffffll : nnzB), cols, topw, rec, recpre, row_begin, tmp, cnt);
    else
        hipLaunchKernelGGL(k_rank_rows<false>, dim3(nrows), dim3(kRankThreads), bytes, s, ab, Bcol,
                           (int)(nnzB > 0x7fffffffll ? 0x7fffffffll : nnzB), cols, topw, rec, recpre, row_begin, tmp, cnt);
    return hipGetLastError();
}

hipError_t launch_dense_rows(int bin, const int2 *ab, const int *Bcol, long long nnzB, int cols,
                             const RowRec *rec, const long long *recpre, int nrows, int row_begin,
                             int *tmp, int *cnt, hipStream_t s)
{
    if (bin == kRankBin) return launch_rank_rows(ab, Bcol, nnzB, cols, rec, recpre, nrows, row_begin, tmp, cnt, s);
    const bool mid = bin == kMidBin;
    if (mid)
        return launch_dense_impl<false, kDenseThreadsMid>(ab, Bcol, nnzB, cols, rec, recpre, nrows, row_begin, tmp, cnt, nullptr, nullptr, s);
    return launch_dense_impl<false, kDenseThreadsBig>(ab, Bcol, nnzB, cols, rec, recpre, nrows, row_begin, tmp, cnt, nullptr, nullptr, s);
}

hipError_t launch_dense_rows_masked(const int2 *ab, const int *Bcol, long long nnzB, int cols,
                                    const RowRec *rec, const long long *recpre, int nrows, int row_begin,
                                    int *tmp, int *cnt, const int *Frow, const int *Fcol, hipStream_t s)
{
    return launch_dense_impl<true, kDenseThreadsBig>(ab, Bcol, nnzB, cols, rec, recpre, nrows, row_begin, tmp, cnt, Frow, Fcol, s);
}

// ---------------------------------------------------------------------------------------
// Exact flow: heavy rows keep the upper-bound placement -- k_dense_rows accumulates AND reads out in the
// symbolic phase (its window bitmap is the expensive part; counting alone would cost almost the same), into
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

// ---------------------------------------------------------------------------------------
// Compaction: every row was written at its upper-bound offset Fprefix[r]; now that the counts
// are scanned into C.row_ptr the rows are copied to their final place.  Pure streaming copy
// (4 B read + 4 B written per output nonzero), driven by the DESTINATION: a workgroup owns
// 4096 consecutive output nonzeros (16 KiB of C.col_idx), takes the rows that cover them from the
// table the count scan left (chunk_row: the row of every 4096th output; without the table -- small
// products -- 32768 outputs or fewer and a binary search in C.row_ptr), keeps their (row_ptr, shift) pairs in LDS 256 rows at a time and
// copies 16 B per lane whenever four outputs lie in one row -- stores are always 16-B aligned
// and fully coalesced, loads are the same stream displaced by the row's shift.  Work per
// workgroup is fixed whatever the row lengths (hub rows and empty rows cost nothing extra).
constexpr int kCompactChunk = 32768;     // output nonzeros per workgroup when its rows are searched (a small product gets smaller chunks: see launch_compact)
constexpr int kCompactChunkTable = 4096; // ... when the count scan left the row table (chunk_row): a multiple of kCompactGran
static_assert(kCompactChunkTable % kCompactGran == 0, "chunk starts must be entries of the row table");
constexpr int kCompactBatch = 256;       // rows staged in LDS at a time
constexpr int kCompactInFlight = 4;      // 16-B groups a thread has in flight (8 measured slower)
constexpr int kCompactSparseRows = 4096; // a chunk spanning more rows than this is searched per output

typedef int v4i __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_compact(const int *__restrict__ tmp,
                                                 const long long *__restrict__ Fprefix,
                                                 const long long *__restrict__ row_ptr,
                                                 int row_lo, int row_hi, int chunk, int *__restrict__ col_idx,
                                                 const int *__restrict__ chunk_row)
{
    __shared__ long long rp[kCompactBatch + 1];
    __shared__ long long sh[kCompactBatch];      // source offset - destination offset of the row
    __shared__ int r_first, r_last;
    const int tid = threadIdx.x;
    const long long out_lo = row_ptr[row_lo], out_hi = row_ptr[row_hi];
    // chunk starts are multiples of 4 outputs so that the 16-B stores stay aligned
    long long o0 = (out_lo & ~3ll) + (long long)blockIdx.x * chunk;
    long long o1 = o0 + chunk;
    if (o0 < out_lo) o0 = out_lo;
    if (o1 > out_hi) o1 = out_hi;
    if (o0 >= o1) return;                        // uniform: the grid is sized by an upper bound
    int rf, rl;                                  // first / last row with outputs in the chunk (rl may be one row further)
    if (chunk_row) {
        // the count scan left the row of every kCompactGran-th output (the chunk is a multiple of that, row_lo == 0):
        // two loads at uniform addresses, no search, no barrier
        rf = chunk_row[o0 / kCompactGran];
        rl = chunk_row[(o1 + kCompactGran - 1) / kCompactGran];       // row of output o1, or of the last output
    } else {
        if (tid == 0) {
            // last row r in [row_lo,row_hi) with row_ptr[r] <= o0: non-empty and contains output o0
            int lo = row_lo, hi = row_hi;            // invariant: row_ptr[lo] <= o0 < row_ptr[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (row_ptr[mid] <= o0) lo = mid; else hi = mid;
            }
            r_first = lo;
            // ... and the row that holds the chunk's last output
            lo = r_first, hi = row_hi;               // invariant: row_ptr[lo] <= o1 - 1 < row_ptr[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (row_ptr[mid] <= o1 - 1) lo = mid; else hi = mid;
            }
            r_last = lo;
        }
        __syncthreads();
        rf = r_first;
        rl = r_last;
    }
    if (chunk_row) {
        // Nearly every chunk of a product with few repeated columns has ONE shift (98 % of the bench matrix's rows have no hole
        // behind them; a row's shift is the holes before it, so first == last means all the same): a plain displaced copy, no row
        // staging, no search, no barrier.  (rl may be the row after the chunk's last: then a hole in between only sends the chunk
        // down the general path.)
        const long long sf = Fprefix[rf] - row_ptr[rf], sl = Fprefix[rl] - row_ptr[rl];
        if (sf == sl) {                              // (uniform)
            const int *__restrict__ src = tmp + sf;
            const int n = (int)(o1 - o0);
            constexpr int kU = 4;
            for (int g0 = tid; 4 * g0 < n; g0 += 256 * kU) {
                Int4U v[kU];
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    const int e = 4 * (g0 + 256 * u);
                    if (e + 3 < n) {                                                          // source only dword aligned
                        const int *q = src + o0 + e;
                        v[u].x = __builtin_nontemporal_load(q), v[u].y = __builtin_nontemporal_load(q + 1);
                        v[u].z = __builtin_nontemporal_load(q + 2), v[u].w = __builtin_nontemporal_load(q + 3);
                    }
                }
#pragma unroll
                for (int u = 0; u < kU; u++) {
                    const int e = 4 * (g0 + 256 * u);
                    if (e + 3 < n) {
                        const v4i w4 = {v[u].x, v[u].y, v[u].z, v[u].w};
                        __builtin_nontemporal_store(w4, reinterpret_cast<v4i *>(col_idx + o0 + e));
                    } else {
                        for (int k = e; k < n; k++) col_idx[o0 + k] = src[o0 + k];             // the product's last outputs
                    }
                }
            }
            return;
        }
    }
    if (rl - rf > kCompactSparseRows) {
        // Mostly empty rows (a masked product, a very sparse result): staging every row of the span
        // through LDS would walk millions of empty rows in ONE workgroup.  Search per output instead.
        for (long long o = o0 + tid; o < o1; o += 256) {
            int lo = rf, hi = rl + 1;            // row_ptr[lo] <= o < row_ptr[hi]
            while (hi - lo > 1) {
                const int mid = (lo + hi) >> 1;
                if (row_ptr[mid] <= o) lo = mid; else hi = mid;
            }
            col_idx[o] = tmp[Fprefix[lo] + (o - row_ptr[lo])];
        }
        return;
    }
    // From here on positions are ints RELATIVE to the chunk's (unclipped, 16-B aligned) start: row starts are clamped to
    // [0, chunk] (a row that begins before the chunk compares like 0, one that begins after it like `chunk`), the row's
    // shift carries the chunk start, so that a group's source is one 64-bit add.
    const long long obase = (out_lo & ~3ll) + (long long)blockIdx.x * chunk;
    const int o0r = (int)(o0 - obase), o1r = (int)(o1 - obase);
    int *rpr = reinterpret_cast<int *>(rp);      // rp's storage, as ints
    int *__restrict__ dst = col_idx + obase;
    int rbase = rf;
    while (true) {
        // rows staged: up to the chunk's last row (row_ptr[rl + 1] >= o1 ends the loop below), a batch at a time
        const int nb = (rl + 1 - rbase < kCompactBatch) ? rl + 1 - rbase : kCompactBatch;
        __syncthreads();
        for (int t = tid; t <= nb; t += 256) {                   // both loads of a row in one round trip
            const long long start = row_ptr[rbase + t];
            const long long rel = start - obase;
            rpr[t] = rel < 0 ? 0 : (rel > chunk ? chunk : (int)rel);
            if (t < nb) sh[t] = Fprefix[rbase + t] - start + obase;
        }
        __syncthreads();
        const int b0 = rpr[0] > o0r ? rpr[0] : o0r;            // outputs covered by this batch and chunk
        const int b1 = rpr[nb] < o1r ? rpr[nb] : o1r;
        // kCompactInFlight 16-B groups per thread per step: independent row searches and loads in flight
        for (int g0 = (b0 >> 2) + tid; (g0 << 2) < b1; g0 += 256 * kCompactInFlight) {
            int o[kCompactInFlight], lo_r[kCompactInFlight];
            long long src[kCompactInFlight];
            bool fast[kCompactInFlight], live[kCompactInFlight];
#pragma unroll
            for (int u = 0; u < kCompactInFlight; u++) {
                o[u] = (g0 + u * 256) << 2;
                live[u] = o[u] < b1;
                const int oo = !live[u] ? b0 : (o[u] > b0 ? o[u] : b0);
                int lo = 0, hi = nb;             // rpr[lo] <= oo < rpr[hi]
                while (hi - lo > 1) {
                    const int mid = (lo + hi) >> 1;
                    if (rpr[mid] <= oo) lo = mid; else hi = mid;
                }
                lo_r[u] = lo;
                fast[u] = live[u] && o[u] >= b0 && o[u] + 3 < b1 && o[u] + 3 < rpr[lo + 1];
                src[u] = o[u] + sh[lo];
            }
            Int4U v[kCompactInFlight];
#pragma unroll
            for (int u = 0; u < kCompactInFlight; u++)
                if (fast[u]) v[u] = *reinterpret_cast<const Int4U *>(tmp + src[u]);     // source only dword aligned
#pragma unroll
            for (int u = 0; u < kCompactInFlight; u++) {
                if (fast[u]) {
                    const v4i w4 = {v[u].x, v[u].y, v[u].z, v[u].w};
                    __builtin_nontemporal_store(w4, reinterpret_cast<v4i *>(dst + o[u]));
                } else if (live[u]) {
                    // a group that straddles rows (or the batch / chunk end): its outputs one by one -- the four loads
                    // first, then the stores (one round trip: nearly every wave has such a group)
                    int r = lo_r[u];
                    int val[4];
                    bool has[4];
#pragma unroll
                    for (int e = 0; e < 4; e++) {
                        const int oe = o[u] + e;
                        has[e] = oe >= b0 && oe < b1;
                        if (has[e]) {
                            while (rpr[r + 1] <= oe) r++;
                            val[e] = tmp[oe + sh[r]];
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; e++)
                        if (has[e]) dst[o[u] + e] = val[e];
                }
            }
        }
        if (rpr[nb] >= o1r || rbase + nb >= row_hi) break;     // uniform: every thread reads the same LDS
        rbase += nb;
    }
}

void launch_compact(const int *tmp, const long long *Fprefix, const long long *row_ptr,
                    int row_lo, int row_hi, long long max_out, int *col_idx, hipStream_t s, const int *chunk_row)
{
    if (row_hi <= row_lo || max_out <= 0) return;
    // chunk: a multiple of 4 outputs (aligned 16-B stores).  With the scan's row table a workgroup's set-up is two
    // loads instead of two binary searches in C.row_ptr, and smaller chunks pay (stitch phase on the bench matrix:
    // 2.35 ms searched at 32768 outputs per workgroup; with the table 2.24 at 32768, 2.18 at 16384, 2.16 at 8192; with
    // the lighter prologue of the final kernel 2.10 at 8192 and 2.11 at 4096, power-law 2.85 -> 2.69 -> 2.43:
    // `profiles/r03_ab_compaction.log`): 4096 outputs (16 KiB).  A small product is
    // cut finer still so that it spreads over the chip (one large chunk would be ONE workgroup walking every row);
    // those chunks are not multiples of the table's grain and are searched.
    long long chunk = ((max_out / 2048) + 3) & ~3ll;
    if (chunk < 256) chunk = 256;
    if (chunk >= kCompactChunkTable && chunk_row && row_lo == 0) {
        chunk = kCompactChunkTable;
    } else {
        chunk_row = nullptr;
        if (chunk > kCompactChunk) chunk = kCompactChunk;
    }
    const int grid = (int)((max_out + 3 + chunk - 1) / chunk) + 1;
    hipLaunchKernelGGL(k_compact, dim3(grid), dim3(256), 0, s, tmp, Fprefix, row_ptr, row_lo, row_hi, (int)chunk, col_idx, chunk_row);
}

}  // namespace bsp
