// capi.hip -- error plumbing, version, compositing dispatch and the small unfused ops.
#include "gs_common.h"
#include "rasterize_common.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

static thread_local char g_err[512] = "";

void gs_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int32_t gs_version(void) { return GS_ABI_VERSION; }
#ifndef GS_HEADER_HASH
#error "GS_HEADER_HASH must be defined by the build (first 8 bytes of sha256(include/gsplat_hip.h), see the Makefile)"
#endif
extern "C" uint64_t gs_header_hash(void) { return GS_HEADER_HASH; }
extern "C" const char *gs_last_error(void) { return g_err; }

static int32_t check_raster_args(const RasterArgs &a) {
    if (a.channels == 0 || a.channels > 513) {
        gs_set_error("rasterize: unsupported number of colour channels: %u", a.channels);
        return 1;
    }
    if (a.tile_size == 0 || a.tile_size > 16) {
        gs_set_error("rasterize: tile_size must be in [1, 16], got %u", a.tile_size);
        return 1;
    }
    if ((uint64_t)a.tile_width * a.tile_size < a.image_width || (uint64_t)a.tile_height * a.tile_size < a.image_height) {
        gs_set_error("rasterize: tile grid %ux%u (tile %u) does not cover the %ux%u image", a.tile_width,
                     a.tile_height, a.tile_size, a.image_width, a.image_height);
        return 1;
    }
    return 0;
}

extern "C" int32_t gs_rasterize_plan(uint32_t n_tiles_all, uint32_t n_isects, uint32_t channels, const int32_t *tuning,
                                     gs_raster_plan *plan) {
    GS_CHECK_ARG(plan != nullptr, "null plan");
    GS_CHECK_ARG(channels >= 1 && channels <= 513, "unsupported number of colour channels");
    return raster_make_plan(n_tiles_all, n_isects, channels, tuning, plan);
}

// the four per-splat arrays: dense rows (NULL strides) or explicit row strides; detects the splat-row form
static int32_t set_splat_layout(RasterArgs &a, const uint32_t *strides) {
    a.s_xy = 2u; a.s_conic = 3u; a.s_color = a.channels; a.s_opac = 1u;
    a.row16 = 0u;
    if (strides != nullptr) {
        a.s_xy = strides[0]; a.s_conic = strides[1]; a.s_color = strides[2]; a.s_opac = strides[3];
        if (a.s_xy < 2u || a.s_conic < 3u || a.s_color < a.channels || a.s_opac < 1u || (a.s_xy & 1u)) {
            gs_set_error("rasterize: splat_strides (%u, %u, %u, %u) too small for rows of 2 / 3 / %u / 1 floats (means2d rows must stay 8-byte aligned)",
                         a.s_xy, a.s_conic, a.s_color, a.s_opac, a.channels);
            return 1;
        }
        a.row16 = (a.channels <= 4u && a.s_xy == 16u && a.s_conic == 16u && a.s_color == 16u && a.s_opac == 16u && a.means2d != nullptr &&
                   ((uintptr_t)a.means2d % 64u) == 0u && a.conics == a.means2d + GS_ROW_CONIC && a.opacities == a.means2d + GS_ROW_OPACITY &&
                   a.colors == a.means2d + GS_ROW_COLOR) ? 1u : 0u;
        // more than 4 channels: the geometry alone from the row (two 16-byte loads of one line), colours from their own array
        if (a.channels > 4u && a.s_xy == 16u && a.s_conic == 16u && a.s_opac == 16u && a.means2d != nullptr && ((uintptr_t)a.means2d % 64u) == 0u &&
            a.conics == a.means2d + GS_ROW_CONIC && a.opacities == a.means2d + GS_ROW_OPACITY)
            a.row16 = 2u;
    }
    return 0;
}

extern "C" int32_t gs_rasterize_fwd(
    uint32_t C, uint32_t n_elems, uint32_t n_isects, uint32_t channels, const float *means2d,
    const float *conics, const float *colors, const float *opacities, const uint32_t *splat_strides, const float *backgrounds,
    const uint8_t *masks, uint32_t image_width, uint32_t image_height, uint32_t tile_size,
    uint32_t tile_width, uint32_t tile_height, const int32_t *tile_offsets,
    const int32_t *flatten_ids, float *render_colors, float *render_alphas, int32_t *last_ids,
    const gs_raster_plan *plan, void *scratch, void *zero_fill, size_t zero_fill_bytes, gs_stream_t stream) {
    GS_CHECK_ARG(render_colors && render_alphas && last_ids && tile_offsets, "null pointer");
    GS_CHECK_ARG(zero_fill_bytes == 0 || (zero_fill && (uintptr_t)zero_fill % 16 == 0 && zero_fill_bytes % 16 == 0),
                 "zero_fill must be 16-byte aligned and a multiple of 16 bytes long");
    GS_CHECK_ARG(n_isects == 0 || (means2d && conics && colors && opacities && flatten_ids), "null pointer");
    GS_CHECK_ARG((plan == nullptr) == (scratch == nullptr), "plan and scratch go together (both or neither)");
    RasterArgs a = {C, n_elems, n_isects, channels, means2d, conics, colors, opacities, backgrounds, masks,
                    image_width, image_height, tile_size, tile_width, tile_height, tile_offsets, flatten_ids,
                    render_colors, render_alphas, last_ids, 0u, 0u, 0u, 0u, 0u, 0u};
    if (int32_t rc = check_raster_args(a)) return rc;
    if (int32_t rc = set_splat_layout(a, splat_strides)) return rc;
    GS_CHECK_ARG(plan == nullptr || raster_plan_ok(plan, C * tile_width * tile_height, n_isects, channels),
                 "plan was not made by gs_rasterize_plan for this (tile count, n_isects, channels)");
    if (C == 0 || image_width == 0 || image_height == 0) {
        if (zero_fill_bytes > 0 && hipMemsetAsync(zero_fill, 0, zero_fill_bytes, (hipStream_t)stream) != hipSuccess)
            { gs_set_error("rasterize: zero fill failed"); return 1; }
        return 0;
    }
    int32_t rc = raster_wave_fwd(a, plan, scratch, zero_fill, zero_fill_bytes, (hipStream_t)stream);
    if (rc) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_rasterize_bwd(
    uint32_t C, uint32_t n_elems, uint32_t n_isects, uint32_t channels, const float *means2d,
    const float *conics, const float *colors, const float *opacities, const uint32_t *splat_strides, const float *backgrounds,
    const uint8_t *masks, uint32_t image_width, uint32_t image_height, uint32_t tile_size,
    uint32_t tile_width, uint32_t tile_height, const int32_t *tile_offsets,
    const int32_t *flatten_ids, const float *render_colors, const float *render_alphas,
    const int32_t *last_ids, const float *v_render_colors, const float *v_render_alphas,
    int64_t v_render_colors_pixel_stride, int64_t v_render_colors_channel_stride, float *v_means2d_abs,
    float *v_means2d, float *v_conics, float *v_colors, float *v_opacities, int32_t packed16, int64_t *det_accum,
    const gs_raster_plan *plan, void *scratch, gs_stream_t stream) {
    GS_CHECK_ARG(det_accum == nullptr || channels <= 4, "the deterministic backward needs channels <= 4");
    GS_CHECK_ARG(v_render_colors_pixel_stride >= 0 && v_render_colors_channel_stride >= 0, "negative gradient stride");
    GS_CHECK_ARG(render_alphas && last_ids && v_render_colors && tile_offsets, "null pointer");
    GS_CHECK_ARG(n_isects == 0 || (means2d && conics && colors && opacities && flatten_ids && v_means2d),
                 "null pointer");
    GS_CHECK_ARG(n_isects == 0 || packed16 || (v_conics && v_colors && v_opacities), "null pointer");
    GS_CHECK_ARG(packed16 >= 0 && packed16 <= 2, "packed16 must be 0, 1 or 2");
    GS_CHECK_ARG(packed16 != 1 || channels <= 4, "packed16 = 1 (every gradient in the rows) needs channels <= 4");
    GS_CHECK_ARG(packed16 != 2 || channels > 4, "packed16 = 2 (geometry rows + separate colour gradients) is the form for more than 4 channels");
    GS_CHECK_ARG(packed16 != 2 || n_isects == 0 || v_colors != nullptr, "packed16 = 2 takes the colour gradients in v_colors");
    GS_CHECK_ARG((plan == nullptr) == (scratch == nullptr), "plan and scratch go together (both or neither)");
    RasterArgs a = {C, n_elems, n_isects, channels, means2d, conics, colors, opacities, backgrounds, masks,
                    image_width, image_height, tile_size, tile_width, tile_height, tile_offsets, flatten_ids,
                    nullptr, nullptr, nullptr, 0u, 0u, 0u, 0u, 0u, 0u};
    RasterGradArgs ga = {render_alphas, last_ids, v_render_colors, v_render_alphas, v_means2d_abs,
                         v_means2d, v_conics, v_colors, v_opacities, 2u, 2u, 3u, channels, 1u, 0u,
                         v_render_colors_pixel_stride, v_render_colors_channel_stride, (long long *)det_accum};
    if (packed16) {
        float *P = v_means2d; // [n_elems,16]: vx vy | ca cb cc | o | c0 c1 c2 c3 | ax ay | pad
        ga.v_means2d = P;
        ga.v_conics = P + 2;
        ga.v_opacities = P + 5;
        ga.v_means2d_abs = v_means2d_abs != nullptr ? P + 10 : nullptr;
        ga.s_abs = ga.s_xy = ga.s_conic = ga.s_opac = 16u;
        if (packed16 == 1) {
            ga.v_colors = P + 6;
            ga.s_color = 16u;
        }
        ga.packed = 1u;
    }
    if (int32_t rc = check_raster_args(a)) return rc;
    if (int32_t rc = set_splat_layout(a, splat_strides)) return rc;
    GS_CHECK_ARG(plan == nullptr || raster_plan_ok(plan, C * tile_width * tile_height, n_isects, channels),
                 "plan was not made by gs_rasterize_plan for this (tile count, n_isects, channels)");
    if (C == 0 || image_width == 0 || image_height == 0 || n_isects == 0) return 0;
    int32_t rc = raster_wave_bwd(a, ga, render_colors, plan, scratch, (hipStream_t)stream);
    if (rc) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// quat/scale -> covariance / precision (unfused public op)
// reference: gsplat/cuda/csrc/quat_scale_to_covar_preci_fwd.cu:19-91, _bwd.cu:20-113
// ---------------------------------------------------------------------------
namespace {

GS_DEV void write_sym(float *dst, const Sym3 &S, int triu, size_t n) {
    if (triu) {
        float *o = dst + 6 * n;
        o[0] = S.xx; o[1] = S.xy; o[2] = S.xz; o[3] = S.yy; o[4] = S.yz; o[5] = S.zz;
    } else {
        float *o = dst + 9 * n;
        o[0] = S.xx; o[1] = S.xy; o[2] = S.xz;
        o[3] = S.xy; o[4] = S.yy; o[5] = S.yz;
        o[6] = S.xz; o[7] = S.yz; o[8] = S.zz;
    }
}

__global__ void __launch_bounds__(GS_BLOCK) qs2cp_fwd_kernel(
    uint32_t N, const float *__restrict__ quats, const float *__restrict__ scales, int triu,
    float *__restrict__ covars, float *__restrict__ precis) {
    uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (n >= N) return;
    const float *q = quats + 4 * (size_t)n;
    const float *s = scales + 3 * (size_t)n;
    Mat3 R = quat_to_rotmat(q[0], q[1], q[2], q[3]);
    if (covars != nullptr) write_sym(covars, covar_from_rot_scale(R, s[0], s[1], s[2]), triu, n);
    if (precis != nullptr) write_sym(precis, covar_from_rot_scale(R, 1.f / s[0], 1.f / s[1], 1.f / s[2]), triu, n);
}

GS_DEV Mat3 read_grad(const float *src, int triu, size_t n) {
    Mat3 G;
    if (triu) {
        // d/d(triu entry): the off-diagonal entry stands for both (i,j) and (j,i)
        const float *v = src + 6 * n;
        G.m[0][0] = v[0]; G.m[0][1] = 0.5f * v[1]; G.m[0][2] = 0.5f * v[2];
        G.m[1][0] = 0.5f * v[1]; G.m[1][1] = v[3]; G.m[1][2] = 0.5f * v[4];
        G.m[2][0] = 0.5f * v[2]; G.m[2][1] = 0.5f * v[4]; G.m[2][2] = v[5];
    } else {
        const float *v = src + 9 * n;
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) G.m[i][j] = v[3 * i + j];
    }
    return G;
}

__global__ void __launch_bounds__(GS_BLOCK) qs2cp_bwd_kernel(
    uint32_t N, const float *__restrict__ quats, const float *__restrict__ scales, int triu,
    const float *__restrict__ v_covars, const float *__restrict__ v_precis,
    float *__restrict__ v_quats, float *__restrict__ v_scales) {
    uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (n >= N) return;
    const float *q = quats + 4 * (size_t)n;
    const float *s = scales + 3 * (size_t)n;
    Mat3 R = quat_to_rotmat(q[0], q[1], q[2], q[3]);
    float vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
    if (v_covars != nullptr)
        covar_vjp_quat_scale(q[0], q[1], q[2], q[3], s[0], s[1], s[2], R, read_grad(v_covars, triu, n), vq, vs);
    if (v_precis != nullptr) {
        // precision = covariance built from 1/s: chain rule d(1/s)/ds = -1/s^2
        float is0 = 1.f / s[0], is1 = 1.f / s[1], is2 = 1.f / s[2];
        float vi[3] = {0.f, 0.f, 0.f};
        covar_vjp_quat_scale(q[0], q[1], q[2], q[3], is0, is1, is2, R, read_grad(v_precis, triu, n), vq, vi);
        vs[0] += -is0 * is0 * vi[0];
        vs[1] += -is1 * is1 * vi[1];
        vs[2] += -is2 * is2 * vi[2];
    }
    float *oq = v_quats + 4 * (size_t)n;
    float *os = v_scales + 3 * (size_t)n;
    oq[0] = vq[0]; oq[1] = vq[1]; oq[2] = vq[2]; oq[3] = vq[3];
    os[0] = vs[0]; os[1] = vs[1]; os[2] = vs[2];
}

} // namespace

extern "C" int32_t gs_quat_scale_to_covar_preci_fwd(
    uint32_t N, const float *quats, const float *scales, int32_t triu, float *covars, float *precis,
    gs_stream_t stream) {
    if (N == 0) return 0;
    GS_CHECK_ARG(quats && scales, "null pointer");
    hipLaunchKernelGGL(qs2cp_fwd_kernel, dim3(gs_div_up(N, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream, N,
                       quats, scales, triu, covars, precis);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_quat_scale_to_covar_preci_bwd(
    uint32_t N, const float *quats, const float *scales, int32_t triu, const float *v_covars,
    const float *v_precis, float *v_quats, float *v_scales, gs_stream_t stream) {
    if (N == 0) return 0;
    GS_CHECK_ARG(quats && scales && v_quats && v_scales, "null pointer");
    hipLaunchKernelGGL(qs2cp_bwd_kernel, dim3(gs_div_up(N, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream, N,
                       quats, scales, triu, v_covars, v_precis, v_quats, v_scales);
    GS_CHECK_LAUNCH();
    return 0;
}
