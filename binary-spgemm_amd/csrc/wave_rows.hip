// wave_rows.hip -- dispatcher of the one-wave-per-row kernels (bodies: wave_rows.inc, built in
// wave_rows_L1..L5.hip).  Picks the number of 5-bit levels from B's column count.
#include "kernels.hpp"

namespace bsp {

template <int LEVELS>
void launch_wave_levels(int bin, const int2 *ab, const int *Bcol, int topw, const RowRec *rec,
                        const long long *recpre, const long long *row_ptr, int nrows, int row_begin, int *tmp, int *cnt,
                        unsigned *err, hipStream_t s, bool count);
extern template void launch_wave_levels<1>(int, const int2 *, const int *, int, const RowRec *, const long long *, const long long *, int, int, int *, int *, unsigned *, hipStream_t, bool);
extern template void launch_wave_levels<2>(int, const int2 *, const int *, int, const RowRec *, const long long *, const long long *, int, int, int *, int *, unsigned *, hipStream_t, bool);
extern template void launch_wave_levels<3>(int, const int2 *, const int *, int, const RowRec *, const long long *, const long long *, int, int, int *, int *, unsigned *, hipStream_t, bool);
extern template void launch_wave_levels<4>(int, const int2 *, const int *, int, const RowRec *, const long long *, const long long *, int, int, int *, int *, unsigned *, hipStream_t, bool);
extern template void launch_wave_levels<5>(int, const int2 *, const int *, int, const RowRec *, const long long *, const long long *, int, int, int *, int *, unsigned *, hipStream_t, bool);

void launch_wave_rows(int bin, int levels, const int2 *ab, const int *Bcol, int cols,
                      const RowRec *rec, const long long *recpre, const long long *row_ptr, int nrows, int row_begin,
                      int *tmp, int *cnt, unsigned *err, hipStream_t s, bool count_only)
{
    if (nrows <= 0) return;
    // words of the directly addressed top bitmap: ceil(cols / 32^levels) <= kWaveTopWords
    const long long span = 1ll << (5 * levels);
    const int topw = (int)(((long long)cols + span - 1) / span);
    switch (levels) {
    case 1: launch_wave_levels<1>(bin, ab, Bcol, topw, rec, recpre, row_ptr, nrows, row_begin, tmp, cnt, err, s, count_only); break;
    case 2: launch_wave_levels<2>(bin, ab, Bcol, topw, rec, recpre, row_ptr, nrows, row_begin, tmp, cnt, err, s, count_only); break;
    case 3: launch_wave_levels<3>(bin, ab, Bcol, topw, rec, recpre, row_ptr, nrows, row_begin, tmp, cnt, err, s, count_only); break;
    case 4: launch_wave_levels<4>(bin, ab, Bcol, topw, rec, recpre, row_ptr, nrows, row_begin, tmp, cnt, err, s, count_only); break;
    default: launch_wave_levels<5>(bin, ab, Bcol, topw, rec, recpre, row_ptr, nrows, row_begin, tmp, cnt, err, s, count_only); break;
    }
}

}  // namespace bsp
