// tile_geometry.h — where a column-major result is cut for the tiled downloads of the SpMM exports (csrc/api.hip): groups
// of whole columns = contiguous pieces of the caller's matrix, cut at page boundaries of the caller's buffer so that no
// page belongs to two registrations.  Plain C++ (no HIP): also compiled and property-tested on the CPU
// (tests/test_tile_geometry.py).
//
//   group g   = columns [gcut[g], gcut[g+1])
//   piece g   = bytes   [hb[g], hb[g+1]) of the result: the page that holds the group's first byte up to (not including) the
//               page that holds the next group's first byte; piece 0 starts at byte 0, the last piece ends at c_bytes
//   tail[g]   = elements at the END of the group's last column that lie in piece g + 1 (the group's last partial page):
//               rows [m - tail[g], m) of column gcut[g+1] - 1.  Tiles of all other rows and columns of group g lie inside
//               piece g.
// ok = false: no tiling (fewer than two groups, an empty piece or group, a tail that is not a whole number of elements or
// does not fit inside the last row block, a leading dimension other than m).
#pragma once
#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <vector>

namespace mx {

struct TileGeometry {
    bool ok = false;
    int ng = 0;
    std::vector<int> gcut, tail;
    std::vector<size_t> hb;
};

inline TileGeometry tile_geometry(uintptr_t c_host, size_t item, int m, int n, size_t ldc, int max_groups, size_t group_bytes,
                                  int last_block_rows)
{
    TileGeometry t;
    const size_t c_bytes = item * (size_t)n * ldc;
    t.ng = (int)std::min<size_t>({(size_t)max_groups, (size_t)std::max(n, 0), c_bytes / group_bytes});
    const int ng = t.ng;
    t.gcut.assign((size_t)std::max(ng, 0) + 1, 0);
    t.tail.assign((size_t)std::max(ng, 1), 0);
    t.hb.assign((size_t)std::max(ng, 0) + 1, 0);
    if (ng < 2 || ldc != (size_t)m) return t;
    for (int g = 0; g <= ng; g++) t.gcut[g] = (int)((int64_t)n * g / ng);
    for (int g = 1; g < ng; g++) {
        const uintptr_t start = c_host + (uintptr_t)((size_t)t.gcut[g] * ldc * item), page = start & ~(uintptr_t)4095;
        const size_t off = page > c_host ? (size_t)(page - c_host) : 0;
        t.hb[g] = std::max(t.hb[g - 1], std::min(off, c_bytes));
    }
    t.hb[ng] = c_bytes;
    for (int g = 0; g < ng; g++) {
        if (t.hb[g + 1] == t.hb[g] || t.gcut[g + 1] == t.gcut[g]) return t;
        if (g + 1 < ng) {
            const size_t end_of_group = (size_t)t.gcut[g + 1] * ldc * item;
            const size_t over = end_of_group > t.hb[g + 1] ? end_of_group - t.hb[g + 1] : 0;
            if (over % item) return t;
            t.tail[g] = (int)(over / item);
            if (t.tail[g] >= last_block_rows) return t;
        }
    }
    t.ok = true;
    return t;
}

}  // namespace mx
