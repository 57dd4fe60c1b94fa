// rasterize_dev.h -- device helpers shared by the compositing kernels of rasterize.hip (1..4 channels, generic route) and
// rasterize_wide.hip (5..32 channels): splat fetch, exact rectangle culling, XCD remap, tile geometry.
#pragma once

#include "gs_common.h"
#include "rasterize_common.h"

namespace {


constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr float ALPHA_MIN = 1.f / 255.f;
constexpr float LOG2_255 = 7.994353436858858f;
constexpr int HEAVY_TILE = 1024; // list length from which a wave raises its priority
// cost classes of the backward's work items: a segment of `seg` entries costs at most 4 * seg record evaluations
constexpr int COST_CLASSES = 32;
GS_DEV uint32_t cost_class(uint32_t c, int32_t seg) { return min(c * 8u / (uint32_t)seg, (uint32_t)COST_CLASSES - 1u); }

struct SplatRaw {
    int32_t g;
    float mx, my, ca, cb, cc, opac;
};

// One splat of the sorted list.  Row form (a.row16, include/gsplat_hip.h "splat rows"): the whole splat is ONE 64-byte line,
// fetched with two (three with the colours) 16-byte loads; otherwise the reference's four arrays through their row strides.
template <int CDIM>
GS_DEV void fetch_splat(const RasterArgs &a, int32_t g, SplatRaw &s, float *col) {
    s.g = g;
    if (a.row16) {
        const float4 *r = reinterpret_cast<const float4 *>(a.means2d + (size_t)g * 16u);
        const float4 r0 = r[0], r1 = r[1];
        s.mx = r0.x; s.my = r0.y; s.ca = r0.z; s.cb = r0.w;
        s.cc = r1.x; s.opac = r1.y;
        if (CDIM > 0) col[0] = r1.z;
        if (CDIM > 1) col[CDIM > 1 ? 1 : 0] = r1.w;
        if (CDIM == 3) col[CDIM > 2 ? 2 : 0] = reinterpret_cast<const float *>(r + 2)[0];
        if (CDIM > 3) {
            const float2 v = reinterpret_cast<const float2 *>(r + 2)[0];
            col[CDIM > 2 ? 2 : 0] = v.x;
            col[CDIM > 3 ? 3 : 0] = v.y;
        }
    } else {
        const float2 xy = *reinterpret_cast<const float2 *>(a.means2d + (size_t)g * a.s_xy);
        const float *cn = a.conics + (size_t)g * a.s_conic;
        s.mx = xy.x; s.my = xy.y;
        s.ca = cn[0]; s.cb = cn[1]; s.cc = cn[2];
        s.opac = a.opacities[(size_t)g * a.s_opac];
#pragma unroll
        for (int k = 0; k < CDIM; ++k) col[k] = a.colors[(size_t)g * a.s_color + k];
    }
}

// 5..32 channels: the geometry from the splat row (a.row16 == 2: means2d / conics / opacities are columns 0 / 2 / 5 of one
// 64-byte-aligned [n_elems,16] buffer) or through the row strides; the colours [ch_off, ch_off + cnt) of the caller's own
// [n_elems, >= channels] array (zeros behind cnt).
template <int CDIM>
GS_DEV void fetch_splat_wide(const RasterArgs &a, int32_t g, SplatRaw &s, float *col, uint32_t ch_off, uint32_t cnt) {
    s.g = g;
    if (a.row16) {
        const float4 *r = reinterpret_cast<const float4 *>(a.means2d + (size_t)g * 16u);
        const float4 r0 = r[0];
        const float2 r1 = reinterpret_cast<const float2 *>(r + 1)[0];
        s.mx = r0.x; s.my = r0.y; s.ca = r0.z; s.cb = r0.w;
        s.cc = r1.x; s.opac = r1.y;
    } else {
        const float2 xy = *reinterpret_cast<const float2 *>(a.means2d + (size_t)g * a.s_xy);
        const float *cn = a.conics + (size_t)g * a.s_conic;
        s.mx = xy.x; s.my = xy.y;
        s.ca = cn[0]; s.cb = cn[1]; s.cc = cn[2];
        s.opac = a.opacities[(size_t)g * a.s_opac];
    }
    const float *cp = a.colors + (size_t)g * a.s_color + ch_off;
    if (cnt == (uint32_t)CDIM) {
        // the whole instance: 16-byte loads at 4-byte alignment (global memory takes dword-aligned multi-dword accesses) -- a
        // 9-float row is two of them and a dword instead of nine dword gathers of 64 different cache lines each (the texture
        // path services one line per lane and instruction: at 16 channels the scalar form cost ~100 us of the forward)
        typedef float v4a4 __attribute__((ext_vector_type(4), aligned(4)));
#pragma unroll
        for (int k = 0; k + 4 <= CDIM; k += 4) {
            const v4a4 v = *reinterpret_cast<const v4a4 *>(cp + k);
            col[k] = v.x; col[k + 1 < CDIM ? k + 1 : 0] = v.y; col[k + 2 < CDIM ? k + 2 : 0] = v.z; col[k + 3 < CDIM ? k + 3 : 0] = v.w;
        }
#pragma unroll
        for (int k = CDIM & ~3; k < CDIM; ++k) col[k] = cp[k];
    } else {
#pragma unroll
        for (int k = 0; k < CDIM; ++k) col[k] = (uint32_t)k < cnt ? cp[k] : 0.f;
    }
}

GS_DEV SplatRaw gather_splat(const RasterArgs &a, int32_t idx, bool in_range) {
    SplatRaw s;
    s.g = 0;
    s.mx = s.my = s.ca = s.cb = s.cc = 0.f;
    s.opac = 0.f;
    if (in_range) {
        float none[1];
        fetch_splat<0>(a, a.flatten_ids[idx], s, none);
    }
    return s;
}

// Exact culling of a splat against a rectangle of pixel centres: the splat can reach
// alpha >= 1/255 somewhere in the rectangle iff min over the rectangle of
// sigma(d) = a/2 dx^2 + b dx dy + c/2 dy^2 is <= ln(255 o).  sigma is convex, so its minimum
// over the box is at the centre (inside: 0) or on one of the (at most two) edges FACING the
// centre -- from the true minimiser the segment towards the centre must leave the box at once --
// and along an edge it is a clamped 1-D parabola.  About 25 VALU per (splat, rectangle), evaluated
// once by the staging lane; the 3-sigma bounding boxes of the tile lists pass ~4x more
// (splat, quadrant) pairs than this test and the axis-aligned extent test ~1.35x more (measured).
struct CullSplat {
    float ha, hc, nbc, nba, t; // a/2, c/2, -b/c, -b/a, threshold with safety margin
    bool pd;                   // positive-definite conic (otherwise: never cull)
};

// false when the splat cannot contribute anywhere (opacity below 1/255, zero, negative or NaN)
GS_DEV bool cull_prepare(const SplatRaw &s, CullSplat &c) {
    c.ha = 0.5f * s.ca;
    c.hc = 0.5f * s.cc;
    c.nbc = c.nba = 0.f;
    c.t = 0.f;
    c.pd = false;
    if (!(s.opac > 0.f)) return false;
    float t = (__log2f(s.opac) + LOG2_255) * LN2; // ln(255 o): alpha >= 1/255 <=> sigma <= t
    if (!(t > -1e-3f)) return false;
    c.t = t * 1.001f + 2e-3f;
    const float det = s.ca * s.cc - s.cb * s.cb;
    c.pd = det > 0.f && s.ca > 0.f && s.cc > 0.f;
    if (c.pd) {
        c.nbc = -s.cb * __builtin_amdgcn_rcpf(s.cc);
        c.nba = -s.cb * __builtin_amdgcn_rcpf(s.ca);
    }
    return true;
}

GS_DEV bool rect_touch(const SplatRaw &s, const CullSplat &c, float x0, float x1, float y0, float y1) {
    const float X0 = x0 - s.mx, X1 = x1 - s.mx, Y0 = y0 - s.my, Y1 = y1 - s.my;
    const float xe = __builtin_amdgcn_fmed3f(0.f, X0, X1), ye = __builtin_amdgcn_fmed3f(0.f, Y0, Y1); // nearest point
    const float dyA = __builtin_amdgcn_fmed3f(c.nbc * xe, Y0, Y1); // best point of the line x = xe
    const float sA = xe * (c.ha * xe + s.cb * dyA) + c.hc * dyA * dyA;
    const float dxB = __builtin_amdgcn_fmed3f(c.nba * ye, X0, X1); // best point of the line y = ye
    const float sB = dxB * (c.ha * dxB + s.cb * ye) + c.hc * ye * ye;
    // margin: the products above cancel for strongly elongated splats; scale the slack with them
    const float slack = 1e-5f * (fabsf(c.ha * xe * xe) + fabsf(c.hc * ye * ye));
    return !c.pd || (fminf(sA, sB) <= c.t + slack);
}

// XCD-aware work-item remap (MI355X: 8 XCDs, each with a private 4 MiB L2; workgroup b runs on
// XCD b % 8).  Consecutive virtual items (neighbouring tiles, which share most of their splats)
// are given to the SAME XCD, so their gathers hit that XCD's L2 instead of re-fetching the
// splat from the fabric on all 8.  Bijective for any M.  Placement only affects speed.
// `group` > 0: XCD x owns every 8th group of `group` consecutive virtual items (locality inside a
// group, load spread over the whole image: the heavy tiles are spatially clustered, so giving one
// XCD a contiguous 1/8 of the image costs more in imbalance than it saves in traffic -- measured).
GS_DEV uint32_t xcd_remap(uint32_t b, uint32_t M, uint32_t group) {
    group &= 0x7fffffffu;
    if (group == 0u) return b;
    const uint32_t full = (M / (8u * group)) * (8u * group); // items covered by complete rounds
    if (b >= full) return b;                                  // ragged tail: identity
    const uint32_t x = b & 7u, i = b >> 3;                    // i-th item of XCD x
    return ((i / group) * 8u + x) * group + (i % group);
}

// Bounding rectangle (pixel centres) of the lanes set in `m` inside an 8x8 quadrant whose first pixel centre is
// (X0, Y0); lane = ly * 8 + lx.  Wave-uniform (scalar bit operations).  m != 0.
struct LiveRect {
    float x0, x1, y0, y1;
};
GS_DEV LiveRect live_rect(unsigned long long m, float X0, float Y0) {
    const uint32_t ylo = (uint32_t)__builtin_ctzll(m) >> 3, yhi = (63u - (uint32_t)__builtin_clzll(m)) >> 3;
    uint32_t c = (uint32_t)m | (uint32_t)(m >> 32);
    c |= c >> 16;
    c |= c >> 8;
    c &= 0xffu; // columns in use
    const uint32_t xlo = (uint32_t)__builtin_ctz(c), xhi = 31u - (uint32_t)__builtin_clz(c);
    LiveRect r;
    r.x0 = X0 + (float)xlo;
    r.x1 = X0 + (float)xhi;
    r.y0 = Y0 + (float)ylo;
    r.y1 = Y0 + (float)yhi;
    return r;
}

struct TileGeom {
    uint32_t lin, cam, tile_id;
    int32_t range_start, range_end;
    uint32_t px0, py0;
};

GS_DEV TileGeom tile_geom(const RasterArgs &a, uint32_t slot) {
    TileGeom g;
    const uint32_t tiles = a.tile_width * a.tile_height;
    g.lin = slot;
    g.cam = g.lin / tiles;
    g.tile_id = g.lin % tiles;
    g.range_start = a.tile_offsets[g.lin];
    g.range_end = (g.lin + 1 == a.C * tiles) ? (int32_t)a.n_isects : a.tile_offsets[g.lin + 1];
    g.px0 = (g.tile_id % a.tile_width) * a.tile_size;
    g.py0 = (g.tile_id / a.tile_width) * a.tile_size;
    return g;
}

// Pixel rectangle (centres) covered by this wave: the tile (NQ == 4) or one quadrant,
// clipped to the tile size and the image.  Wave-uniform.
struct Rect {
    float x0, x1, y0, y1;
    bool empty;
};

template <int NQ>
GS_DEV Rect wave_rect(const RasterArgs &a, const TileGeom &tg, uint32_t q_first) {
    uint32_t ox0 = (NQ == 4) ? 0u : 8u * (q_first & 1u), oy0 = (NQ == 4) ? 0u : 8u * (q_first >> 1);
    uint32_t ox1 = (NQ == 4) ? 16u : ox0 + 8u, oy1 = (NQ == 4) ? 16u : oy0 + 8u;
    ox1 = min(ox1, a.tile_size);
    oy1 = min(oy1, a.tile_size);
    uint32_t X0 = tg.px0 + ox0, Y0 = tg.py0 + oy0;
    uint32_t X1 = min(tg.px0 + ox1, a.image_width), Y1 = min(tg.py0 + oy1, a.image_height);
    Rect r;
    r.empty = (ox0 >= ox1) || (oy0 >= oy1) || X0 >= X1 || Y0 >= Y1;
    r.x0 = (float)X0 + 0.5f;
    r.y0 = (float)Y0 + 0.5f;
    r.x1 = (float)X1 - 0.5f;
    r.y1 = (float)Y1 - 0.5f;
    return r;
}

// the segmented backward's work list (built by seg_items_build_kernel inside gs_rasterize_fwd) and the forward's checkpoints
struct SegArgs {
    const uint2 *items;          // [COST_CLASSES][max_items] (tile, k), one region per cost class
    const uint32_t *class_count; // [COST_CLASSES] items per class
    uint32_t max_items;          // region length
    const float *ckpt;           // [k][CDIM+1][256]
    const float *render_colors;
    int32_t seg;
};

} // namespace
