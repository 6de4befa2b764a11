// Property test of csrc/tile_geometry.h on the CPU (tests/test_tile_geometry.py): for random result geometries, every byte a
// tile's copies write lies inside ONE registered piece — the piece of its column group, or the next one for the group's
// last partial page —, interior piece boundaries are page boundaries of the caller's buffer, and the tiles cover the
// result exactly once.  The copies are those of the `tile` lambda in csrc/api.hip: a pitched copy of the group's whole
// columns over the block's rows; in the LAST row block the group's last column is split at the page boundary.
#include "../../matrixextra_amd/csrc/tile_geometry.h"
#include <cstdio>
#include <cstdlib>
#include <random>

int main(int argc, char **argv)
{
    const int cases = argc > 1 ? atoi(argv[1]) : 20000;
    std::mt19937_64 rng(12345);
    long tiled = 0, checked_tiles = 0;
    for (int c = 0; c < cases; c++) {
        const size_t item = rng() % 2 ? 4 : 8;
        const int n = 1 + (int)(rng() % 300);
        // small "pages worth" of rows so that many cases tile: the geometry only looks at bytes, scale group_bytes down with it
        const int m = 1024 + (int)(rng() % 200000);
        const size_t ldc = rng() % 20 == 0 ? (size_t)m + 8 : (size_t)m;
        const size_t group_bytes = (size_t)1 << (14 + rng() % 8);                 // 16 KiB .. 2 MiB (64 MiB in the library)
        const uintptr_t base = ((uintptr_t)1 << 40) + (uintptr_t)(rng() % 4096) / item * item;
        const int nblk = 2 + (int)(rng() % 15);
        std::vector<int> cut((size_t)nblk + 1);
        for (int b = 0; b <= nblk; b++) cut[b] = b == nblk ? m : (int)((int64_t)m * b / nblk) & ~1023;
        const mx::TileGeometry t = mx::tile_geometry(base, item, m, n, ldc, 16, group_bytes, cut[nblk] - cut[nblk - 1]);
        const size_t c_bytes = item * (size_t)n * ldc;
        if (!t.ok) {
            if (ldc == (size_t)m && t.ng >= 2) {
                // refused although groups exist: one of the documented reasons must hold
                bool reason = false;
                for (int g = 0; g < t.ng; g++) {
                    if (t.hb[g + 1] == t.hb[g] || t.gcut[g + 1] == t.gcut[g]) reason = true;
                    if (g + 1 < t.ng && t.tail[g] >= cut[nblk] - cut[nblk - 1]) reason = true;
                }
                // (the element-size remainder cannot happen with an item-aligned base)
                if (!reason) { printf("case %d: refused without a reason\n", c); return 1; }
            }
            continue;
        }
        tiled++;
        const int ng = t.ng;
        if (t.hb[0] != 0 || t.hb[ng] != c_bytes) { printf("case %d: ends\n", c); return 1; }
        for (int g = 1; g < ng; g++)
            if (t.hb[g] <= t.hb[g - 1] || (base + t.hb[g]) % 4096) { printf("case %d: piece %d not on a page boundary\n", c, g); return 1; }
        size_t covered = 0;
        auto inside = [&](size_t lo, size_t hi, int piece) { return lo >= t.hb[piece] && hi <= t.hb[piece + 1]; };
        for (int g = 0; g < ng; g++) {
            for (int b = 0; b < nblk; b++) {
                const int c0 = cut[b], c1 = cut[b + 1], g0 = t.gcut[g], g1 = t.gcut[g + 1];
                if (c1 == c0) continue;
                checked_tiles++;
                const int tl = b == nblk - 1 && g + 1 < ng ? t.tail[g] : 0;
                const int whole_cols = tl ? g1 - g0 - 1 : g1 - g0;
                for (int col = g0; col < g0 + whole_cols; col++) {                 // the pitched copy, column by column
                    const size_t lo = ((size_t)col * ldc + (size_t)c0) * item, hi = lo + (size_t)(c1 - c0) * item;
                    if (!inside(lo, hi, g)) { printf("case %d: tile (%d, %d) column %d leaves piece %d\n", c, b, g, col, g); return 1; }
                    covered += hi - lo;
                }
                if (tl) {
                    const size_t lo = ((size_t)(g1 - 1) * ldc + (size_t)c0) * item, head = (size_t)(c1 - c0 - tl) * item;
                    if (head && !inside(lo, lo + head, g)) { printf("case %d: head of the last column of group %d leaves its piece\n", c, g); return 1; }
                    if (!inside(lo + head, lo + head + (size_t)tl * item, g + 1)) { printf("case %d: fragment of group %d not in piece %d\n", c, g, g + 1); return 1; }
                    if (lo + head != t.hb[g + 1]) { printf("case %d: the split of group %d is not at the piece boundary\n", c, g); return 1; }
                    covered += (size_t)(c1 - c0) * item;
                }
            }
        }
        if (covered != c_bytes) { printf("case %d: covered %zu of %zu bytes\n", c, covered, c_bytes); return 1; }
    }
    printf("tile geometry ok: %ld of %d cases tiled, %ld tiles checked\n", tiled, cases, checked_tiles);
    return tiled > cases / 20 ? 0 : 2;
}
