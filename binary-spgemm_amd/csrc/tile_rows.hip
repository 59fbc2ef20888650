// tile_rows.hip -- host side of the fused flow's kernel (body: tile_rows.inc): picks the number of
// 5-bit levels of the tile accumulator from the key width, sizes the persistent grid.
#include "tile_rows.inc"

namespace bsp {

static int ceil_log2_ll(long long x)
{
    int b = 0;
    while (b < 62 && (1ll << b) < x) b++;
    return b;
}

// A tile's accumulator holds keys (local row << col_bits) | column.  With L levels it resolves
// 16 + 5 (L - 1) key bits (a top bitmap of 2048 words = 2^16 bits, then 5 bits per level), so the
// rows a tile may hold follow from the level count: row_bits = min(6, resolved - col_bits).  The
// smallest L whose tiles can hold the expected number of rows is taken (every level costs two LDS
// reads, an atomic and a scan per product).
int tile_levels_for(int cols, long long est_rows_per_tile, int cap, int *row_bits, int *col_bits)
{
    const int top_bits = 5 + ceil_log2_ll(cap);            // the top bitmap has as many words as a tile has products
    int cb = ceil_log2_ll(cols > 1 ? cols : 1);
    if (cb < 5) cb = 5;                                // a level-0 slot (32 keys) never spans two rows
    if (est_rows_per_tile < 1) est_rows_per_tile = 1;
    if (est_rows_per_tile > 64) est_rows_per_tile = 64;
    const int want = ceil_log2_ll(est_rows_per_tile);
    int best_l = 5, best_rb = 0;
    for (int l = 1; l <= 5; l++) {
        const int resolved = top_bits + 5 * (l - 1);
        if (resolved < cb) continue;
        int rb = resolved - cb;
        if (rb > 6) rb = 6;
        if (rb > 32 - cb) rb = 32 - cb;
        if (rb < 0) rb = 0;
        best_l = l;
        best_rb = rb;
        if (rb >= want) break;
    }
    *row_bits = best_rb;
    *col_bits = cb;
    return best_l;
}

hipError_t launch_tile_rows(int levels, const TileArgs &a, int grid, hipStream_t s)
{
    if (grid <= 0 || a.ntiles <= 0) return hipSuccess;
    switch (levels) {
    case 1: return launch_tile_levels<1>(a, grid, s);
    case 2: return launch_tile_levels<2>(a, grid, s);
    case 3: return launch_tile_levels<3>(a, grid, s);
    case 4: return launch_tile_levels<4>(a, grid, s);
    case 5: return launch_tile_levels<5>(a, grid, s);
    default: return hipErrorInvalidValue;
    }
}

int tile_rows_grid(int levels, int device)
{
    switch (levels) {
    case 1: return tile_levels_grid<1>(device);
    case 2: return tile_levels_grid<2>(device);
    case 3: return tile_levels_grid<3>(device);
    case 4: return tile_levels_grid<4>(device);
    default: return tile_levels_grid<5>(device);
    }
}

}  // namespace bsp
