// sh.hip -- R2: spherical-harmonics colour evaluation, fwd + bwd (gfx950).
//
// Replaces gsplat/cuda/csrc/compute_sh_fwd.cu:12-72, compute_sh_bwd.cu:14-95 and the
// device functions of gsplat/cuda/include/spherical_harmonics.cuh:13-362.
//
// Design notes (MI355X):
//   * This is the largest HBM term of a degree-3 step (192 B of coefficients per
//     splat).  One lane owns one gaussian and ALL three colour channels, so the
//     coefficient row is read once with 16-byte loads (the reference reads it from
//     three different lanes) and the gradient row is written once with 16-byte stores.
//   * Coefficients shared across cameras are indexed [N,K,3] directly; the reference
//     materialises a [C,N,K,3] copy and autograd then sums a [C,N,K,3] gradient.  Here
//     the lane loops over cameras and keeps the K*3 gradient accumulators in VGPRs.
//   * bwd writes every row of v_coeffs (zeros for culled splats / inactive bands), so
//     the caller never zero-fills 192 B/splat first.
//   * The basis is evaluated through the (x+iy)^m recurrence (fC_m, fS_m) and its
//     gradient uses d fC_m = m (fC_{m-1}, -fS_{m-1}), d fS_m = m (fS_{m-1}, fC_{m-1}).
//   * The contraction is [1 x K] . [K x 3] with both operands unique per splat: no
//     operand reuse, so MFMA cannot beat VALU here; kept on VALU.
#include "gs_common.h"
#include "sh_eval.h"
#include "sh_bwd_lane.h"

namespace {

// SPLIT: the coefficient rows are the pair (coeffs [N,1,3], view.coeffs_rest [N,K-1,3]) -- a template parameter, because a
// run-time choice between the two loaders costs the registers of both (60 -> 113 VGPRs at degree 3)
template <int DEG, bool VEC, bool SPLIT>
__global__ void __launch_bounds__(GS_BLOCK) sh_fwd_kernel(
    uint32_t C, uint32_t N, uint32_t K, const float *__restrict__ dirs,
    const float *__restrict__ coeffs, int shared, const uint8_t *__restrict__ masks,
    float *__restrict__ colors, ShView view) {
    constexpr int NB = ShDim<DEG>::NB;
    uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    uint32_t c = blockIdx.y;
    if (n >= N) return;
    size_t e = (size_t)c * N + n;
    if (view.opac_out != nullptr) view.opac_out[e] = view.opac_in[n]; // every element, visible or not (like .repeat)
    if (!sh_active(masks, view, e)) return;
    float dx = 0.f, dy = 0.f, dz = 1.f;
    if (DEG >= 1) sh_dir(dirs, view, c, n, e, dx, dy, dz);
    const float *row = coeffs + (shared ? (size_t)n : e) * (SPLIT ? 3u : K * 3);
    const float *rest = SPLIT ? view.coeffs_rest + (size_t)n * (K - 1) * 3 : nullptr;
    float r, g, b;
    sh_view_color<DEG, VEC, SPLIT ? 1 : 0>(dx, dy, dz, row, rest, view.clamp_half != 0, r, g, b);
    float *co = colors + e * view.color_stride;
    co[0] = r;
    co[1] = g;
    co[2] = b;
}

// One lane per gaussian; loops over cameras.  SHARED: v_coeffs is [N,K,3] and the lane
// accumulates over cameras in registers; otherwise [C,N,K,3] rows are written per camera.
template <int DEG, bool VEC, bool SHARED>
__global__ void __launch_bounds__(GS_BLOCK) sh_bwd_kernel(
    uint32_t C, uint32_t N, uint32_t K, const float *__restrict__ dirs,
    const float *__restrict__ coeffs, const uint8_t *__restrict__ masks,
    const float *__restrict__ v_colors, float *__restrict__ v_coeffs,
    float *__restrict__ v_dirs, ShView view, const float *__restrict__ colors_out, uint32_t v_colors_stride,
    float *__restrict__ v_means) {
    uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (n >= N) return; // (a partial last wave never takes the staged store paths: they need wave_n0 + 64 <= N)
    float vmx, vmy, vmz;
    bool any_on;
    sh_bwd_lane<DEG, VEC, SHARED>(C, N, K, n, true, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs, view, colors_out, v_colors_stride,
                                  v_dirs != nullptr || v_means != nullptr, vmx, vmy, vmz, any_on);
    if (v_means != nullptr && (any_on || !view.prefilled)) {
        v_means[3 * (size_t)n] = vmx; v_means[3 * (size_t)n + 1] = vmy; v_means[3 * (size_t)n + 2] = vmz;
    }
}

bool rows_vectorizable(const void *p, uint32_t K) { return ((uintptr_t)p % 16 == 0) && ((K * 3) % 4 == 0); }

template <int DEG>
void launch_fwd(bool vec, dim3 grid, hipStream_t st, uint32_t C, uint32_t N, uint32_t K, const float *dirs,
                const float *coeffs, int shared, const uint8_t *masks, float *colors, ShView view) {
    if (view.coeffs_rest != nullptr)
        hipLaunchKernelGGL((sh_fwd_kernel<DEG, false, true>), grid, dim3(GS_BLOCK), 0, st, C, N, K, dirs, coeffs, shared, masks, colors, view);
    else if (vec)
        hipLaunchKernelGGL((sh_fwd_kernel<DEG, true, false>), grid, dim3(GS_BLOCK), 0, st, C, N, K, dirs, coeffs, shared, masks, colors, view);
    else
        hipLaunchKernelGGL((sh_fwd_kernel<DEG, false, false>), grid, dim3(GS_BLOCK), 0, st, C, N, K, dirs, coeffs, shared, masks, colors, view);
}

template <int DEG>
void launch_bwd(bool vec, bool shared, dim3 grid, hipStream_t st, uint32_t C, uint32_t N, uint32_t K,
                const float *dirs, const float *coeffs, const uint8_t *masks, const float *v_colors,
                float *v_coeffs, float *v_dirs, ShView view, const float *colors_out, uint32_t vstride, float *v_means) {
#define GS_SH_BWD(V, S)                                                                               \
    hipLaunchKernelGGL((sh_bwd_kernel<DEG, V, S>), grid, dim3(GS_BLOCK), 0, st, C, N, K, dirs, coeffs, \
                       masks, v_colors, v_coeffs, v_dirs, view, colors_out, vstride, v_means)
    if (vec && shared) GS_SH_BWD(true, true);
    else if (vec) GS_SH_BWD(true, false);
    else if (shared) GS_SH_BWD(false, true);
    else GS_SH_BWD(false, false);
#undef GS_SH_BWD
}

} // namespace

extern "C" int32_t gs_sh_fwd(
    uint32_t C, uint32_t N, uint32_t K, uint32_t degree, const float *dirs, const float *coeffs,
    int32_t coeffs_shared, const uint8_t *masks, float *colors, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(coeffs && colors, "null pointer");
    GS_CHECK_ARG(degree <= 4, "degree must be <= 4");
    GS_CHECK_ARG((degree + 1) * (degree + 1) <= K, "K too small for degree");
    GS_CHECK_ARG(degree == 0 || dirs != nullptr, "dirs required for degree >= 1");
    dim3 grid(gs_div_up(N, GS_BLOCK), C);
    hipStream_t st = (hipStream_t)stream;
    bool vec = rows_vectorizable(coeffs, K);
    ShView view = {nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0u, nullptr, nullptr, nullptr, 3u, 0};
    switch (degree) {
        case 0: launch_fwd<0>(vec, grid, st, C, N, K, dirs, coeffs, coeffs_shared, masks, colors, view); break;
        case 1: launch_fwd<1>(vec, grid, st, C, N, K, dirs, coeffs, coeffs_shared, masks, colors, view); break;
        case 2: launch_fwd<2>(vec, grid, st, C, N, K, dirs, coeffs, coeffs_shared, masks, colors, view); break;
        case 3: launch_fwd<3>(vec, grid, st, C, N, K, dirs, coeffs, coeffs_shared, masks, colors, view); break;
        default: launch_fwd<4>(vec, grid, st, C, N, K, dirs, coeffs, coeffs_shared, masks, colors, view); break;
    }
    GS_CHECK_LAUNCH();
    return 0;
}

// camera centres: -A^-1 t of the affine world->camera matrix [[A, t], [0, 1]] (adjugate form)
__global__ void camera_centers_kernel(uint32_t C, const float *__restrict__ viewmats, float *__restrict__ out) {
    uint32_t c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    camera_center(viewmats + 16 * c, out[3 * c], out[3 * c + 1], out[3 * c + 2]);
}

extern "C" int32_t gs_camera_centers(uint32_t C, const float *viewmats, float *campos, gs_stream_t stream) {
    if (C == 0) return 0;
    GS_CHECK_ARG(viewmats && campos, "null pointer");
    hipLaunchKernelGGL(camera_centers_kernel, dim3(gs_div_up(C, 64)), dim3(64), 0, (hipStream_t)stream, C, viewmats, campos);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_sh_view_fwd(
    uint32_t C, uint32_t N, uint32_t K, uint32_t degree, const float *means, const float *campos, int32_t campos_from_viewmats,
    const float *coeffs, const float *coeffs_rest, const int32_t *radii, float *colors, uint32_t colors_stride, const float *opacities,
    float *opacities_cn, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(means && campos && coeffs && colors, "null pointer");
    GS_CHECK_ARG(colors_stride >= 3, "colors_stride must be >= 3");
    GS_CHECK_ARG(degree <= 4, "degree must be <= 4");
    GS_CHECK_ARG((degree + 1) * (degree + 1) <= K, "K too small for degree");
    dim3 grid(gs_div_up(N, GS_BLOCK), C);
    hipStream_t st = (hipStream_t)stream;
    bool vec = coeffs_rest != nullptr ? (K * 3) % 4 == 0 : rows_vectorizable(coeffs, K);
    GS_CHECK_ARG(coeffs_rest == nullptr || K >= 2, "split coefficients need K >= 2");
    ShView view = {means, campos, radii, 1, campos_from_viewmats, opacities, opacities_cn, nullptr, 0u, nullptr, coeffs_rest, nullptr, colors_stride, 0};
    GS_CHECK_ARG((opacities == nullptr) == (opacities_cn == nullptr), "opacities and opacities_cn go together");
    switch (degree) {
        case 0: launch_fwd<0>(vec, grid, st, C, N, K, nullptr, coeffs, 1, nullptr, colors, view); break;
        case 1: launch_fwd<1>(vec, grid, st, C, N, K, nullptr, coeffs, 1, nullptr, colors, view); break;
        case 2: launch_fwd<2>(vec, grid, st, C, N, K, nullptr, coeffs, 1, nullptr, colors, view); break;
        case 3: launch_fwd<3>(vec, grid, st, C, N, K, nullptr, coeffs, 1, nullptr, colors, view); break;
        default: launch_fwd<4>(vec, grid, st, C, N, K, nullptr, coeffs, 1, nullptr, colors, view); break;
    }
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_sh_bwd(
    uint32_t C, uint32_t N, uint32_t K, uint32_t degree, const float *dirs, const float *coeffs,
    int32_t coeffs_shared, const uint8_t *masks, const float *v_colors, float *v_coeffs,
    float *v_dirs, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(coeffs && v_colors && v_coeffs, "null pointer");
    GS_CHECK_ARG(degree <= 4, "degree must be <= 4");
    GS_CHECK_ARG((degree + 1) * (degree + 1) <= K, "K too small for degree");
    GS_CHECK_ARG(degree == 0 || dirs != nullptr, "dirs required for degree >= 1");
    dim3 grid(gs_div_up(N, GS_BLOCK));
    hipStream_t st = (hipStream_t)stream;
    bool vec = rows_vectorizable(coeffs, K) && rows_vectorizable(v_coeffs, K);
    bool shared = coeffs_shared != 0;
    ShView view = {nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, nullptr, 0u, nullptr, nullptr, nullptr, 3u, 0};
    switch (degree) {
        case 0: launch_bwd<0>(vec, shared, grid, st, C, N, K, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs, view, nullptr, 3, nullptr); break;
        case 1: launch_bwd<1>(vec, shared, grid, st, C, N, K, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs, view, nullptr, 3, nullptr); break;
        case 2: launch_bwd<2>(vec, shared, grid, st, C, N, K, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs, view, nullptr, 3, nullptr); break;
        case 3: launch_bwd<3>(vec, shared, grid, st, C, N, K, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs, view, nullptr, 3, nullptr); break;
        default: launch_bwd<4>(vec, shared, grid, st, C, N, K, dirs, coeffs, masks, v_colors, v_coeffs, v_dirs, view, nullptr, 3, nullptr); break;
    }
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_sh_view_bwd(
    uint32_t C, uint32_t N, uint32_t K, uint32_t degree, const float *means, const float *campos, int32_t campos_from_viewmats,
    const float *coeffs, const float *coeffs_rest, const int32_t *radii, const float *colors_out, uint32_t colors_out_stride,
    const float *v_colors, uint32_t v_colors_stride, float *v_coeffs, float *v_coeffs_rest, float *v_means, const float *v_opacities_cn, uint32_t v_opacities_stride,
    float *v_opacities, int32_t outputs_prefilled, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(means && campos && coeffs && colors_out && v_colors && v_coeffs, "null pointer");
    GS_CHECK_ARG(degree <= 4, "degree must be <= 4");
    GS_CHECK_ARG((degree + 1) * (degree + 1) <= K, "K too small for degree");
    GS_CHECK_ARG(v_colors_stride >= 3 && colors_out_stride >= 3, "colour row strides must be >= 3");
    dim3 grid(gs_div_up(N, GS_BLOCK));
    hipStream_t st = (hipStream_t)stream;
    bool vec = rows_vectorizable(coeffs, K) && rows_vectorizable(v_coeffs, K);
    GS_CHECK_ARG((coeffs_rest == nullptr) == (v_coeffs_rest == nullptr), "coeffs_rest and v_coeffs_rest go together");
    if (coeffs_rest != nullptr) { // split rows: the staged wave stores need 16-byte aligned gradient tensors and 3K % 4 == 0
        GS_CHECK_ARG(K >= 2, "split coefficients need K >= 2");
        vec = (K * 3) % 4 == 0 && (uintptr_t)v_coeffs % 16 == 0 && (uintptr_t)v_coeffs_rest % 16 == 0;
    }
    ShView view = {means, campos, radii, 1, campos_from_viewmats, nullptr, nullptr, v_opacities_cn, v_opacities_stride, v_opacities, coeffs_rest, v_coeffs_rest, colors_out_stride, outputs_prefilled != 0};
    GS_CHECK_ARG((v_opacities_cn == nullptr) == (v_opacities == nullptr), "v_opacities_cn and v_opacities go together");
    switch (degree) {
        case 0: launch_bwd<0>(vec, true, grid, st, C, N, K, nullptr, coeffs, nullptr, v_colors, v_coeffs, nullptr, view, colors_out, v_colors_stride, v_means); break;
        case 1: launch_bwd<1>(vec, true, grid, st, C, N, K, nullptr, coeffs, nullptr, v_colors, v_coeffs, nullptr, view, colors_out, v_colors_stride, v_means); break;
        case 2: launch_bwd<2>(vec, true, grid, st, C, N, K, nullptr, coeffs, nullptr, v_colors, v_coeffs, nullptr, view, colors_out, v_colors_stride, v_means); break;
        case 3: launch_bwd<3>(vec, true, grid, st, C, N, K, nullptr, coeffs, nullptr, v_colors, v_coeffs, nullptr, view, colors_out, v_colors_stride, v_means); break;
        default: launch_bwd<4>(vec, true, grid, st, C, N, K, nullptr, coeffs, nullptr, v_colors, v_coeffs, nullptr, view, colors_out, v_colors_stride, v_means); break;
    }
    GS_CHECK_LAUNCH();
    return 0;
}
