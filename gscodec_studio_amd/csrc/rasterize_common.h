// rasterize_common.h -- argument blocks and wave64 primitives shared by the compositing kernels.
#pragma once

#include "gs_common.h"

struct RasterArgs {
    uint32_t C, n_elems, n_isects, channels;
    const float *means2d;
    const float *conics;
    const float *colors;
    const float *opacities;
    const float *backgrounds;
    const uint8_t *masks;
    uint32_t image_width, image_height, tile_size, tile_width, tile_height;
    const int32_t *tile_offsets;
    const int32_t *flatten_ids;
    float *render_colors;
    float *render_alphas;
    int32_t *last_ids;
    uint32_t xcd_group; // XCD-aware work-item grouping (0 = off), set by the dispatcher
    // row strides (in floats) of means2d / conics / colors / opacities: 2, 3, channels, 1 for the reference's dense
    // arrays; all 16 when the four pointers are column views of ONE splat row (include/gsplat_hip.h, "splat rows").
    uint32_t s_xy, s_conic, s_color, s_opac;
    uint32_t row16; // 1: the pointers alias one 64-byte-aligned [n_elems,16] buffer at columns 0 / 2 / 6 / 5 and channels <= 4:
                    // the kernels fetch a splat as (up to) three 16-byte loads from ONE 64-byte line
    const uint32_t *tile_order; // forward: workgroup b composites tile tile_order[b] (NULL: the XCD remap of b), set by the dispatcher
};

// a buffer the forward zero-fills on the side: n float4s, per_block of them per tile workgroup
struct ZeroFill {
    float4 *ptr;
    size_t n;
    uint32_t per_block;
};

struct RasterGradArgs {
    const float *render_alphas;
    const int32_t *last_ids;
    const float *v_render_colors;
    const float *v_render_alphas;
    float *v_means2d_abs;
    float *v_means2d;
    float *v_conics;
    float *v_colors;
    float *v_opacities;
    // row strides (in floats) of the five gradient arrays.  Separate tensors: 2, 2, 3, channels, 1.
    // Packed mode (one 64-byte row per splat: [vx vy | ca cb cc | o | c0..c3 | ax ay | pad]): all 16.
    uint32_t s_abs, s_xy, s_conic, s_color, s_opac;
    uint32_t packed; // 1: the pointers above alias one [n_elems,16] buffer
    // element strides of v_render_colors per pixel / per channel: (channels, 1) for a dense [C,H,W,channels] tensor,
    // (0, 0) for the broadcast gradient of sum(render) (autograd hands over an expanded scalar: nothing to materialise)
    int64_t s_vrc_pix, s_vrc_ch;
    // Deterministic mode (opt-in): the per-splat sums are accumulated in FIXED POINT, because integer adds commute: the
    // result no longer depends on the order in which the (tile, segment) work items reach a splat (float atomics make the
    // low bits of every gradient change from run to run; so do the reference's).  int64 [n_elems, 2, 12], columns as in the
    // splat rows; TWO accumulators per value, because the contributions span more than 63 bits of dynamic range (conic
    // gradients of large splats reach 1e11, position gradients of small ones 1e-8):
    //   |v| <  2^10 : bin 0, units of 2^-38 (exact to fp32 precision down to |v| = 2^-15; up to 2^15 contributions fit)
    //   |v| >= 2^10 : bin 1, units of 2^-6  (relative resolution <= 2^-16 per contribution; range 1.4e17)
    // A second kernel converts bin 0 * 2^-38 + bin 1 * 2^-6 into the float outputs.
    long long *det;
};

constexpr float GS_DET_SPLIT = 1024.f;
constexpr float GS_DET_SCALE_LO = 274877906944.f; // 2^38
constexpr float GS_DET_SCALE_HI = 64.f;           // 2^6
constexpr double GS_DET_INV_LO = 1.0 / 274877906944.0, GS_DET_INV_HI = 1.0 / 64.0;

// one gradient contribution: float atomic into `p`, or the fixed-point one into det[row][bin][comp].
// DET: 1 / 0 = decided at compile time (the hot segmented kernel: a run-time choice costs it registers), -1 = by ga.det
template <int DET = -1>
GS_DEV void grad_add(const RasterGradArgs &ga, float *p, size_t row, uint32_t comp, float v) {
    if (DET == 1 || (DET < 0 && ga.det != nullptr)) {
        const bool hi = !(fabsf(v) < GS_DET_SPLIT);
        const long long q = __float2ll_rn(v * (hi ? GS_DET_SCALE_HI : GS_DET_SCALE_LO));
        atomicAdd(reinterpret_cast<unsigned long long *>(ga.det + row * 24u + (hi ? 12u : 0u) + comp), (unsigned long long)q);
    } else {
        unsafeAtomicAdd(p, v);
    }
}

// 64-lane sum with DPP row shifts + row broadcasts (GFX9 family).  The total is valid in
// lane 63 only.  All 64 lanes must be active.
GS_DEV float wave_reduce_sum_dpp(float v) {
    // row_shr:1, row_shr:2, row_shr:4, row_shr:8 -> inclusive scan inside each row of 16
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, false));
    // row_bcast:15 into rows 1 and 3, row_bcast:31 into rows 2 and 3
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));
    return v;
}

int32_t raster_make_plan(uint32_t n_tiles_all, uint32_t n_isects, uint32_t channels, const int32_t *tuning, gs_raster_plan *plan);
bool raster_plan_ok(const gs_raster_plan *plan, uint32_t n_tiles_all, uint32_t n_isects, uint32_t channels);
int32_t raster_wave_fwd(const RasterArgs &a, const gs_raster_plan *plan, void *scratch, void *zero_fill, size_t zero_fill_bytes,
                        hipStream_t st);
int32_t raster_wave_bwd(const RasterArgs &a, const RasterGradArgs &ga, const float *render_colors, const gs_raster_plan *plan,
                        void *scratch, hipStream_t st);
// rasterize_wide.hip: one launch of the segmented backward over channels [ch_off, ch_off + cnt), 5 <= cnt <= 16
void raster_seg_bwd_wide(const RasterArgs &a, const RasterGradArgs &ga, uint32_t max_items, int use_va, const void *items,
                         const uint32_t *class_count, const float *ckpt, const float *render_colors, int32_t seg, uint32_t ch_off,
                         uint32_t cnt, hipStream_t st);
