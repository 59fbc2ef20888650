// kernels.hpp -- launch wrappers of the HIP kernels behind bspgemm_multiply (host-callable).
// Kernel bodies: rowwork.hip, scan.hip, wave_rows.hip, dense_rows.hip, compact.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bsp {

// one-wave-per-row capacity classes: bin b (1..6) holds rows with F_i <= 64 << (b-1)
constexpr int kNumBins = 8;             // 0 empty, 1..6 wave rows, 7 dense-window rows
constexpr int kWaveBins = 6;
constexpr int kMaxWaveCap = 64 << (kWaveBins - 1);   // 2048 products
constexpr int kRowsPerWave = 8;         // consecutive list entries handled by one wave

// number of 6-bit levels below the directly addressed top bitmap for `cols` columns
//   LEVELS=1: cols <= 64^1*64 = 4096   (top bitmap IS the column bitmap)
//   LEVELS=2: cols <= 2^18, LEVELS=3: cols <= 2^24, LEVELS=4: up to 2^31 (128 top words)
inline int levels_for_cols(int64_t cols)
{
    if (cols <= (1ll << 12)) return 1;
    if (cols <= (1ll << 18)) return 2;
    if (cols <= (1ll << 24)) return 3;
    return 4;
}

// F_i = sum_{j in A_i} |B_j| for rows [row_begin,row_end)  ->  F[i-row_begin]
void launch_row_work(const int *Arow, const int *Acol, const int *Brow,
                     int row_begin, int row_end, long long *F, hipStream_t s);

// exclusive scan of F (int64) into prefix[0..n] and, fused, classification of every row into
// the capacity bins: bin_rows[b*n + k] = absolute row id, bin_count[b] (device, zeroed here)
void launch_scan_and_bin(const long long *F, int n, int row_begin, long long *prefix,
                         long long *partials, int *bin_rows, int *bin_count, int *cnt,
                         hipStream_t s);

// exclusive scan of int32 counts into int64 prefix[0..n]
void launch_scan_counts(const int *cnt, int n, long long *prefix, long long *partials, hipStream_t s);

// numeric phase, one wave per row (rank-bitmap accumulator)
void launch_wave_rows(int bin, int levels, const int *Arow, const int *Acol,
                      const int *Brow, const int *Bcol, int cols,
                      const int *rows, int nrows, int row_begin,
                      const long long *Fprefix, int *tmp, int *cnt, hipStream_t s);

// numeric phase, one workgroup per heavy row (windowed dense LDS bitmap)
hipError_t launch_dense_rows(const int *Arow, const int *Acol, const int *Brow, const int *Bcol,
                             int cols, const int *rows, int nrows, int row_begin,
                             const long long *Fprefix, int *tmp, int *cnt, hipStream_t s);

// tmp[Fprefix[r] .. +cnt[r])  ->  col_idx[row_ptr[r] ..)
void launch_compact(const int *tmp, const long long *Fprefix, const long long *row_ptr,
                    int nrows, int *col_idx, hipStream_t s);

// rows longer than 8192 entries (listed in the dense bin) are copied by a workgroup each
void launch_compact_big(const int *tmp, const long long *Fprefix, const long long *row_ptr,
                        const int *rows, int nrows, int row_begin, int *col_idx, hipStream_t s);

// row_ptr rebasing helpers for uploads and the multi-GPU stitch
void launch_rebase_i32(int *row_ptr, int n, int base, hipStream_t s);
void launch_add_base_i64(long long *dst, const long long *src, int n, long long base, hipStream_t s);

}  // namespace bsp
