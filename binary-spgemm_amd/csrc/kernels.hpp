// kernels.hpp -- launch wrappers of the HIP kernels behind bspgemm_multiply (host-callable).
// Kernel bodies: prepass.hip (row work, scans, class records), wave_rows.inc (+ wave_rows_L*.hip),
// wave_masked.hip, dense_rows.hip (heavy rows + compaction).
// Tuning constants are compile-time constants, not switches: what was tried against them is in profiles/.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

namespace bsp {

// capacity classes of a row by its product count F_i:
//   0          : empty
//   1..16      : one wavefront per row, capacity 64*kWaveChunks[b] products.  The kernel body is
//                straight-line over its CHUNKS 64-product chunks (exact s_waitcnt counts, no
//                branches), so a row costs what its CLASS costs: the classes step by one chunk up
//                to 512 products, by two up to 1024, then by four/eight
//   kRankBin   : 2048 < F_i <= rank_cap_for_cols(cols) (0 = no such class for this column count; 6144 for 2^18 < cols <= 2^24):
//                a 512-thread workgroup with a two-level rank bitmap -- LDS and read-out proportional to the row (dense_rows.hip)
//   kMidBin    : dense-window rows, up to mid_cap_for_cols(cols) products: 512-thread workgroups, four per CU
//   kDenseBin  : dense-window rows, above that: one 1024-thread workgroup per row
constexpr int kWaveBins = 16;
constexpr int kNumBins = kWaveBins + 4;
constexpr int kRankBin = kWaveBins + 1;
constexpr int kMidBin = kWaveBins + 2;
constexpr int kDenseBin = kWaveBins + 3;
constexpr int kRankCap = 6144;          // products (= slots) of a rank-class row
int rank_cap_for_cols(long long cols);  // kRankCap, or 0 where the class is not used (dense_rows.hip)
constexpr int kMaxBins = 20;            // size of the per-class arrays in bspgemm_stats
constexpr int kWaveChunks[kWaveBins + 1] = {0, 1, 2, 3, 4, 5, 6, 7, 8, 10, 12, 14, 16, 20, 24, 28, 32};
constexpr int kMaxWaveCap = 64 * kWaveChunks[kWaveBins];   // 2048 products
// products up to which a heavy row takes the small (512-thread) shape: its 32 KiB window covers 2^18
// columns, so the row's products are gathered ceil(cols / 2^18) times -- the fewer passes, the larger
// the rows it is good for.  Swept with BSPGEMM_MID_CAP on the round-2 build (numeric phase, ms):
// Graph500-skew scale 18 (one pass) 131072: 6.47, 524288: 5.87, 4194304: 6.18; power-law n = 2^20 (four
// passes) 8192: 13.18, 32768: 12.88, 131072: 12.59, 262144: 12.96 -- i.e. 524288 / passes in both cases.
constexpr int kMidCap = 524288;
inline int mid_cap_for_cols(long long cols)
{
    const long long passes = (cols + (1ll << 18) - 1) >> 18;
    if (passes <= 1) return kMidCap;
    const long long c = kMidCap / passes;
    return c < kMaxWaveCap ? kMaxWaveCap : (int)c;      // (== kMaxWaveCap: no row takes the small shape)
}
constexpr int kRowsPerWave = 16;        // consecutive list entries handled by one wave

constexpr int kWaveTopWords = 256;      // 32-bit words of the directly addressed top bitmap

// number of 5-bit levels of the rank bitmap for `cols` columns: the smallest L whose top bitmap
// ceil(cols / 32^L) fits kWaveTopWords words
//   L=1: cols <= 8192 (the top bitmap IS the column bitmap)   L=2: <= 2^18   L=3: <= 2^23
//   L=4: <= 2^28                                              L=5: any int32 column count
inline int levels_for_cols(int64_t cols)
{
    for (int L = 1; L < 5; L++)
        if (cols <= ((int64_t)kWaveTopWords << (5 * L))) return L;
    return 5;
}

// k_wave_rows trades a fourth level for a larger top bitmap up to 2^24 columns (512 top words at
// three levels): one ranked level costs two LDS reads, an atomic and a blocked scan per product
// chunk, the larger top only a longer scan/clear per row (measured at 2^24 columns: -6 % kernel
// time; 1024 words at 2^25 columns loses to four levels: too few resident waves).
constexpr int kWaveTopWordsWide = 512;
inline int wave_levels_for_cols(int64_t cols)
{
    const int L = levels_for_cols(cols);
    if (L == 4 && cols <= ((int64_t)kWaveTopWordsWide << 15)) return 3;
    return L;
}

// one record per non-empty row, grouped by capacity class (written by the prepass)
struct RowRec {
    int row;    // absolute row id
    int a0;     // A.row_ptr[row]
    int alen;   // |A_row|
    int f;      // F_row (clamped to INT_MAX)
};

// F_i = sum_{j in A_i} |B_j| for rows [row_begin,row_end)  ->  F[i-row_begin];
// ab[jj] = (start of B row A.col_idx[jj], |B_j|) for every A-nonzero of those rows.  What is gathered per A-nonzero:
//   Bblk8 != NULL   B's blocked extents table (launch_blk8); with Bpad (the padded row_ptr) its bases and the starts are
//                   positions in the PADDED copy of B.col_idx
//   else Bext       the per-row table {start in the padded copy, length}
//   else            a B.row_ptr pair (starts in B.col_idx itself)
void launch_row_work(const int *Arow, const int *Acol, const int *Brow, const int *Bblk8, const int *Bpad, const int2 *Bext,
                     int row_begin, int row_end, long long *F, int2 *ab, hipStream_t s);
// blk[3b..3b+2] = {start_ptr[8b] (NULL: row_ptr), lengths of rows 8b..8b+7 clamped to 255, one byte each}
// *clamped_nnz (device, zeroed by the caller) += nonzeros in rows of 255 or more
void launch_blk8(const int *row_ptr, const int *start_ptr, int n, int *blk, unsigned long long *clamped_nnz, hipStream_t s);
// padded copy of an operand's col_idx (rows on 64-byte boundaries): plen[j] = ceil(len_j / 16) * 16; then, with pad_ptr =
// its exclusive scan, the rows copied to col_pad and (ext != NULL) ext[j] = {pad_ptr[j], len_j}
void launch_pad_lengths(const int *row_ptr, int n, int *plen, hipStream_t s);
void launch_pad_copy(const int *row_ptr, const int *col_idx, const int *pad_ptr, int n, int *col_pad, int2 *ext, hipStream_t s);

// deg8[i] = min(|row i|, 255): the per-operand byte table behind launch_row_products
void launch_deg8(const int *row_ptr, int n, unsigned char *deg8, hipStream_t s);
// F only (no extents), through B's byte table of row lengths
void launch_row_products(const int *Arow, const int *Acol, const int *Brow, const unsigned char *Bdeg8,
                         int row_begin, int row_end, long long *F, hipStream_t s);

// exclusive scan of F (int64) into prefix[0..n] and, fused, classification of every row into
// the capacity classes: rec[] holds the non-empty rows grouped by class (class b starts at
// sum(bin_count[1..b-1])), recpre[] their output offsets; bin_count[8] on the device.
// bound_cols > 0: rows are CLASSIFIED by F_i but PLACED by min(F_i, bound_cols) (prefix[] and recpre[]
// are offsets of that bound: a row cannot have more outputs than B has columns -- what keeps the
// upper-bound workspace of a skewed product near nnz(C) instead of near F).
// heavy_cols > 0: the heavy rows (class kDenseBin) get their own workspace offsets in recpre --
// exclusive prefix of min(F_i, heavy_cols) over the heavy rows -- and hpartials[ceil(n/2048)]
// holds that workspace's total size; one-wave rows are then placed by the symbolic counts.
// `scal` (device, may be NULL): everything the host reads back after the prepass in one struct --
// without a heavy-row workspace (heavy_cols <= 0) also the product count, summed from `true_F` when
// the rows are sized by something else than their products (masked multiply), else from F.
struct PrepScalars {
    long long totalF;
    long long heavy_total;
    long long products;
    int a_lo, a_hi;
    int bin_count[kNumBins];
};
void launch_scan_and_bin(const long long *F, int n, int row_begin, const int *Arow, long long *prefix,
                         long long *partials, int *bin_tiles, int *bin_count, RowRec *rec,
                         long long *recpre, int *cnt, int heavy_cols, long long *hpartials, int mid_cap, int rank_cap, int bound_cols,
                         hipStream_t s, PrepScalars *scal = nullptr, const long long *true_F = nullptr);

// prefix[0..n] = *base + exclusive scan of the int32 counts (base NULL = 0; may alias prefix[0])
// `chunk_row` (may be NULL; needs base == NULL): chunk_row[c] = the row that holds output c * kCompactGran, for every
// such output below the total, and chunk_row[ceil(total / kCompactGran)] = the row of the last output -- the
// compaction's row look-up, left behind by the scan that knows every row's range anyway
constexpr int kCompactGran = 4096;
inline size_t compact_chunk_rows(long long max_out) { return (size_t)((max_out + kCompactGran - 1) / kCompactGran) + 2; }
void launch_scan_counts(const int *cnt, int n, long long *prefix, long long *partials,
                        const long long *base, hipStream_t s, int *chunk_row = nullptr);

// numeric phase, one wave per row (rank-bitmap accumulator).  row_ptr != NULL: row i is written at
// tmp + row_ptr[i - row_begin] (tmp = C.col_idx, sizes known from the symbolic phase, cnt may be
// NULL); row_ptr == NULL: at its upper-bound offset tmp + recpre[k], |C_i| to cnt (masked product)
// count_only: the symbolic twin of the same kernel -- nothing is emitted (tmp, recpre, row_ptr unused), cnt[i] = |C_i|
// err (device, never NULL): bit 0 is set when a row's gathered product count exceeds its class capacity -- impossible
// for consistent operands (the classes come from the same extents), seen only when an operand was rewritten under
// the library; the row is then truncated to its capacity instead of overrunning LDS
constexpr unsigned kErrCapacity = 1u;     // a row gathered more products than its capacity class holds
constexpr unsigned kErrStaleTable = 2u;   // an operand's derived tables do not match its row_ptr (bspgemm_matrix_invalidate)
void launch_wave_rows(int bin, int levels, const int2 *ab, const int *Bcol, int cols,
                      const RowRec *rec, const long long *recpre, const long long *row_ptr, int nrows,
                      int row_begin, int *tmp, int *cnt, unsigned *err, hipStream_t s, bool count_only = false);
// debug check (BSPGEMM_OPT_CHECK): deg8[] / blk8[] / the padded row_ptr (each may be NULL) against row_ptr; sets kErrStaleTable in *err
void launch_check_tables(const int *row_ptr, int rows, const unsigned char *deg8, const int *blk8, const int *pad_ptr, unsigned *err,
                         hipStream_t s);

// heavy rows: workspace -> final place (one workgroup per heavy row)
void launch_place_heavy(const int *tmp, const RowRec *rec, const long long *recpre, int nrows,
                        const long long *row_ptr, int row_begin, int *col_idx, hipStream_t s);

// numeric phase, one workgroup per heavy row (windowed dense LDS bitmap); mid: the 512-thread shape
hipError_t launch_dense_rows(int bin, const int2 *ab, const int *Bcol, long long nnzB, int cols,
                             const RowRec *rec, const long long *recpre, int nrows, int row_begin,
                             int *tmp, int *cnt, hipStream_t s);

// the heavy rows' records in order of decreasing products (n <= kHeavySortMax), into rec_out / pre_out
constexpr int kHeavySortMax = 8192;
void launch_order_heavy(const RowRec *rec, const long long *recpre, int n, RowRec *rec_out, long long *pre_out, hipStream_t s);

// masked variant: C = F .* (A*B); every non-empty row goes through the window kernel, which
// keeps only the product bits that F's row (absolute row id, F.row_ptr/F.col_idx) admits
hipError_t launch_dense_rows_masked(const int2 *ab, const int *Bcol, long long nnzB, int cols,
                                    const RowRec *rec, const long long *recpre, int nrows, int row_begin,
                                    int *tmp, int *cnt, const int *Frow, const int *Fcol, hipStream_t s);

// mask-first one-wave path of the masked product (mask rows <= 2048 entries, cols <= 2^23)
bool wave_masked_supported(int cols);
void launch_wave_masked(int bin, const int2 *ab, const int *Bcol, int cols, const int *Frow, const int *Fcol,
                        const RowRec *rec, const long long *recpre, int nrows, int row_begin,
                        int *tmp, int *cnt, hipStream_t s);
// mlen[i] = |F's row i| when row i has products, else 0: what the masked product bins and offsets by
void launch_mask_lengths(const long long *F, const int *Frow, int row_begin, int n, long long *mlen, hipStream_t s);

// rows [row_lo,row_hi): tmp[Fprefix[r] .. +cnt[r])  ->  col_idx[row_ptr[r] ..).  The output range
// is read from row_ptr on the device; `max_out` (an upper bound of its length, e.g. the rows'
// product count) only sizes the grid.
void launch_compact(const int *tmp, const long long *Fprefix, const long long *row_ptr,
                    int row_lo, int row_hi, long long max_out, int *col_idx, hipStream_t s, const int *chunk_row = nullptr);

// ---- single-round-trip path for small products (small.hip) ----------------------------------------
constexpr int kSmallMaxProducts = 65536;   // products (and entries of C.col_idx / the workspace) the path is sized for
constexpr int kSmallMaxRow = 2048;         // products of one row (one wave sorts them in LDS)
constexpr int kSmallMaxRows = 1 << 17;     // rows multiplied (one workgroup scans them)
constexpr int kSmallMaxNnzA = 32768;       // A-nonzeros in the multiplied rows (host-side eligibility)
struct SmallScalars {                      // what the host reads back, once, at the end
    long long totalF;                      // products
    long long nnzC;
    int bail;                              // 1: the product does not fit this path (nothing was written): take the general flow
    int nonempty;                          // rows with products
    int a_lo, a_hi;                        // A.row_ptr at the ends of the row range
};
struct SmallTiles {                        // device scratch of the path's two scans (per 32 rows / per 256 rows)
    long long f32[kSmallMaxRows / 32];     // products
    int nz32[kSmallMaxRows / 32];          // non-empty rows
    int mx32[kSmallMaxRows / 32];          // largest row
    int c256[kSmallMaxRows / 256];         // outputs
};
// list[] (ints, >= nrows), F / Fprefix (nrows + 1), cnt (nrows), tmp and col_idx (kSmallMaxProducts each), row_ptr (nrows + 1)
void launch_small(const int *Arow, const int *Acol, const int *Brow, const int *Bcol, int row_begin, int nrows,
                  long long *F, long long *Fprefix, int *list, int *cnt, int *tmp, long long *row_ptr, int *col_idx,
                  SmallTiles *tl, SmallScalars *sc, hipStream_t s);

// int64 row_ptr -> int32 (operand form of a product)
void launch_narrow_row_ptr(const long long *src, int *dst, int n, hipStream_t s);
// T = A or I as CSR: row i gets column i appended (duplicates are legal in operands)
void launch_add_diagonal(const int *Arow, const int *Acol, int n, int *Trow, int *Tcol, hipStream_t s);

// row_ptr rebasing helper for interior-pointer uploads
void launch_rebase_i32(int *row_ptr, int n, int base, hipStream_t s);

}  // namespace bsp
