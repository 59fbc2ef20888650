// one translation unit per LEVELS value of the rank-bitmap kernel (see wave_rows.inc)
#include "wave_rows.inc"
namespace bsp {
template void launch_wave_levels<1>(int, const int2 *, const int *, int, const RowRec *, const long long *,
                                    const long long *, int, int, int *, int *, unsigned *, hipStream_t, bool);
}
