// isect_common.h -- the tile box of a projected splat, shared by the binning kernels (isect.hip, bin.hip).
#pragma once

#include "gs_common.h"

struct TileBox {
    int32_t x0, y0, x1, y1; // min inclusive, max exclusive
};

// isect_tiles.cu:56-69.  The reference casts a possibly negative float to uint32 and
// relies on the saturating conversion; here the clamp is explicit.
GS_DEV TileBox tile_box(float mx, float my, int32_t radius, float tile_size, int32_t tw, int32_t th) {
    float tr = (float)radius / tile_size;
    float tx = mx / tile_size;
    float ty = my / tile_size;
    TileBox b;
    b.x0 = min(max(0, (int32_t)floorf(tx - tr)), tw);
    b.y0 = min(max(0, (int32_t)floorf(ty - tr)), th);
    b.x1 = min(max(0, (int32_t)ceilf(tx + tr)), tw);
    b.y1 = min(max(0, (int32_t)ceilf(ty + tr)), th);
    return b;
}

