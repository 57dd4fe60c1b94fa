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

namespace {

// v_n = sum_k w_k * grad Y_k   (w_k = sum_c coeff[k][c] * v_colour[c]); bands >= 1 only.
template <int DEG>
GS_DEV void sh_basis_grad_contract(float x, float y, float z, const float *w, float &gx, float &gy, float &gz) {
    gx = gy = gz = 0.f;
    if (DEG < 1) return;
    const float k1 = 0.48860251190292f;
    gy += -k1 * w[1];
    gz += k1 * w[2];
    gx += -k1 * w[3];
    if (DEG < 2) return;
    float z2 = z * z;
    float c1 = x * x - y * y, s1 = 2.f * x * y;
    const float k2 = 0.5462742152960395f, k2b = -1.092548430592079f;
    gx += k2 * 2.f * y * w[4] + k2b * z * w[7] + k2 * 2.f * x * w[8];
    gy += k2 * 2.f * x * w[4] + k2b * z * w[5] - k2 * 2.f * y * w[8];
    gz += k2b * y * w[5] + 2.f * 0.9461746957575601f * z * w[6] + k2b * x * w[7];
    if (DEG < 3) return;
    float c2 = x * c1 - y * s1, s2 = x * s1 + y * c1;
    float t0c = -2.285228997322329f * z2 + 0.4570457994644658f;
    float t0c_z = -2.f * 2.285228997322329f * z;
    float t1b = 1.445305721320277f * z;
    const float k3 = -0.5900435899266435f;
    gx += k3 * 3.f * s1 * w[9] + t1b * 2.f * y * w[10] + t0c * w[13] + t1b * 2.f * x * w[14] + k3 * 3.f * c1 * w[15];
    gy += k3 * 3.f * c1 * w[9] + t1b * 2.f * x * w[10] + t0c * w[11] - t1b * 2.f * y * w[14] - k3 * 3.f * s1 * w[15];
    float y12_z = 3.f * 1.865881662950577f * z2 - 1.119528997770346f;
    gz += 1.445305721320277f * s1 * w[10] + t0c_z * y * w[11] + y12_z * w[12] + t0c_z * x * w[13] +
          1.445305721320277f * c1 * w[14];
    if (DEG < 4) return;
    float t0d = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    float t0d_z = -3.f * 4.683325804901025f * z2 + 2.007139630671868f;
    float t1c = 3.31161143515146f * z2 - 0.47308734787878f;
    float t1c_z = 2.f * 3.31161143515146f * z;
    float t2b = -1.770130769779931f * z;
    const float k4 = 0.6258357354491763f;
    float y12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
    gx += k4 * 4.f * s2 * w[16] + t2b * 3.f * s1 * w[17] + t1c * 2.f * y * w[18] + t0d * w[21] +
          t1c * 2.f * x * w[22] + t2b * 3.f * c1 * w[23] + k4 * 4.f * c2 * w[24];
    gy += k4 * 4.f * c2 * w[16] + t2b * 3.f * c1 * w[17] + t1c * 2.f * x * w[18] + t0d * w[19] -
          t1c * 2.f * y * w[22] - t2b * 3.f * s1 * w[23] - k4 * 4.f * s2 * w[24];
    float y20_z = 1.984313483298443f * (y12 + z * y12_z) - 1.006230589874905f * 2.f * 0.9461746957575601f * z;
    gz += -1.770130769779931f * s2 * w[17] + t1c_z * s1 * w[18] + t0d_z * y * w[19] + y20_z * w[20] +
          t0d_z * x * w[21] + t1c_z * c1 * w[22] - 1.770130769779931f * c2 * w[23];
}

// store CNT active floats followed by zeros up to row_len
template <int CNT, bool VEC>
GS_DEV void store_row(float *__restrict__ p, const float *src, uint32_t row_len) {
    if (VEC) {
        constexpr int NV = CNT / 4;
#pragma unroll
        for (int i = 0; i < NV; ++i)
            reinterpret_cast<float4 *>(p)[i] = make_float4(src[4 * i], src[4 * i + 1], src[4 * i + 2], src[4 * i + 3]);
        constexpr int REM = CNT - NV * 4;
        uint32_t done = NV * 4;
        if (REM > 0) {
            float t[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < REM; ++i) t[i] = src[NV * 4 + i];
            reinterpret_cast<float4 *>(p)[NV] = make_float4(t[0], t[1], t[2], t[3]);
            done += 4;
        }
        for (uint32_t i = done; i < row_len; i += 4) reinterpret_cast<float4 *>(p + i)[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
#pragma unroll
        for (int i = 0; i < CNT; ++i) p[i] = src[i];
        for (uint32_t i = CNT; i < row_len; ++i) p[i] = 0.f;
    }
}

template <bool VEC>
GS_DEV void zero_row(float *__restrict__ p, uint32_t row_len) {
    if (VEC) {
        for (uint32_t i = 0; i < row_len; i += 4) reinterpret_cast<float4 *>(p + i)[0] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
        for (uint32_t i = 0; i < row_len; ++i) p[i] = 0.f;
    }
}

// "View" mode (used by rasterization()): directions are means[n] - campos[c] computed in-kernel,
// the mask is radii[c,n] > 0 and the output is clamp_min(colour + 0.5, 0) -- i.e. the torch ops
// around the reference's spherical_harmonics call (rendering.py:372-392) are fused in.
struct ShView {
    const float *means;   // [N,3] or nullptr (then `dirs` is used)
    const float *campos;  // [C,3] camera centres, or [C,4,4] world->camera matrices when from_viewmats
    const int32_t *radii; // [C,N] or nullptr
    int clamp_half;       // colour = max(colour + 0.5, 0)
    int from_viewmats;    // the centre is derived in-kernel (wave-uniform math; saves the gs_camera_centers launch)
    // per-view opacities riding along (rendering.py:331 `opacities.repeat(C, 1)` and the sum over cameras in its backward):
    const float *opac_in;    // fwd: [N]
    float *opac_out;         // fwd: [C,N] <- opac_in[n]
    const float *v_opac_cn;  // bwd: [C,N] rows with stride v_opac_stride floats
    uint32_t v_opac_stride;
    float *v_opac_out;       // bwd: [N] <- sum over cameras
    const float *coeffs_rest; // split rows (sh_eval.h): coeffs is [N,1,3], this is [N,K-1,3]; NULL = one [N,K,3] tensor
    float *v_coeffs_rest;     // bwd: gradient of coeffs_rest, [N,K-1,3]
    uint32_t color_stride;   // row stride (floats) of the colours the forward writes / the backward reads back: 3, or 16 when
                             // they are columns of the splat rows (include/gsplat_hip.h)
    int prefilled;           // bwd, shared coefficients: v_coeffs (/ v_coeffs_rest) hold zeros already -- rows of gaussians no
                             // camera sees are not stored, v_means is only written for the others
};

GS_DEV bool sh_active(const uint8_t *masks, const ShView &v, size_t e) {
    if (masks != nullptr) return masks[e] != 0;
    if (v.radii != nullptr) return v.radii[e] > 0;
    return true;
}

GS_DEV void sh_dir(const float *dirs, const ShView &v, uint32_t c, uint32_t n, size_t e, float &dx, float &dy, float &dz) {
    if (v.means != nullptr) {
        float cx, cy, cz;
        if (v.from_viewmats) {
            camera_center(v.campos + 16 * c, cx, cy, cz);
        } else {
            cx = v.campos[3 * c]; cy = v.campos[3 * c + 1]; cz = v.campos[3 * c + 2];
        }
        dx = v.means[3 * (size_t)n] - cx;
        dy = v.means[3 * (size_t)n + 1] - cy;
        dz = v.means[3 * (size_t)n + 2] - cz;
    } else {
        dx = dirs[3 * e]; dy = dirs[3 * e + 1]; dz = dirs[3 * e + 2];
    }
}

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
    constexpr int NB = ShDim<DEG>::NB;
    uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (n >= N) return;
    float vmx = 0.f, vmy = 0.f, vmz = 0.f; // view mode: d/d means = sum over cameras of d/d dirs
    if (view.v_opac_out != nullptr) { // culled (c, n) pairs hold exact zeros in the gradient rows
        float vo = 0.f;
        for (uint32_t c = 0; c < C; ++c) vo += view.v_opac_cn[((size_t)c * N + n) * view.v_opac_stride];
        view.v_opac_out[n] = vo;
    }
    const uint32_t row_len = K * 3;
    const bool split = SHARED && view.coeffs_rest != nullptr; // (uniform)
    float acc[NB * 3];
    if (SHARED) {
#pragma unroll
        for (int i = 0; i < NB * 3; ++i) acc[i] = 0.f;
    }
    float cf[NB * 3];
    bool have_cf = false;
    bool any_on = false;
    for (uint32_t c = 0; c < C; ++c) {
        size_t e = (size_t)c * N + n;
        bool on = sh_active(masks, view, e);
        any_on |= on;
        if (!on) {
            if (!SHARED) zero_row<VEC>(v_coeffs + e * row_len, row_len);
            if (v_dirs != nullptr) {
                v_dirs[3 * e] = 0.f; v_dirs[3 * e + 1] = 0.f; v_dirs[3 * e + 2] = 0.f;
            }
            continue;
        }
        const float *vcp = v_colors + e * v_colors_stride;
        float vr = vcp[0], vg = vcp[1], vb = vcp[2];
        if (view.clamp_half) { // gradient of clamp_min(colour + 0.5, 0): passes where the output is > 0
            const float *co = colors_out + e * view.color_stride;
            if (!(co[0] > 0.f)) vr = 0.f;
            if (!(co[1] > 0.f)) vg = 0.f;
            if (!(co[2] > 0.f)) vb = 0.f;
        }
        float Y[NB];
        float x = 0.f, y = 0.f, z = 1.f, inv = 1.f;
        if (DEG >= 1) {
            float dx, dy, dz;
            sh_dir(dirs, view, c, n, e, dx, dy, dz);
            inv = rsqrtf(dx * dx + dy * dy + dz * dz);
            x = dx * inv; y = dy * inv; z = dz * inv;
        }
        sh_basis<DEG>(x, y, z, Y);
        if (SHARED) {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                acc[3 * k] += Y[k] * vr;
                acc[3 * k + 1] += Y[k] * vg;
                acc[3 * k + 2] += Y[k] * vb;
            }
        } else {
            float out[NB * 3];
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                out[3 * k] = Y[k] * vr;
                out[3 * k + 1] = Y[k] * vg;
                out[3 * k + 2] = Y[k] * vb;
            }
            store_row<NB * 3, VEC>(v_coeffs + e * row_len, out, row_len);
        }
        if (v_dirs != nullptr || v_means != nullptr) {
            float gx = 0.f, gy = 0.f, gz = 0.f;
            if (DEG >= 1) {
                if (!SHARED || !have_cf) {
                    if (split) load_coeff_row<NB * 3, VEC>(coeffs + 3 * (size_t)n, view.coeffs_rest + (size_t)n * (K - 1) * 3, cf);
                    else load_floats<NB * 3, VEC>(coeffs + (SHARED ? (size_t)n : e) * row_len, cf);
                    have_cf = true;
                }
                float w[NB];
#pragma unroll
                for (int k = 0; k < NB; ++k) w[k] = cf[3 * k] * vr + cf[3 * k + 1] * vg + cf[3 * k + 2] * vb;
                float vx, vy, vz;
                sh_basis_grad_contract<DEG>(x, y, z, w, vx, vy, vz);
                float dot = vx * x + vy * y + vz * z;
                gx = (vx - dot * x) * inv;
                gy = (vy - dot * y) * inv;
                gz = (vz - dot * z) * inv;
            }
            if (v_dirs != nullptr) {
                v_dirs[3 * e] = gx; v_dirs[3 * e + 1] = gy; v_dirs[3 * e + 2] = gz;
            }
            vmx += gx; vmy += gy; vmz += gz;
        }
    }
    if (SHARED) {
        // Every lane holds one 4*NV-float gradient row; rows of neighbouring lanes are row_len floats apart, so
        // storing them straight from registers puts the 64 lanes of each store on 64 different cache lines.
        // When the rows are dense (row_len == NB*3, multiple of 4) a full wave transposes them through LDS
        // and writes its 64 rows as one contiguous block, 1 KiB per store instruction.
        constexpr int RL = NB * 3;
        constexpr bool CAN_T = VEC && (RL % 4 == 0);
        __shared__ float4 s_tr[CAN_T ? (GS_BLOCK / GS_WAVE) * GS_WAVE * (RL / 4) : 1];
        const uint32_t lane = threadIdx.x % GS_WAVE, wave = threadIdx.x / GS_WAVE;
        const uint32_t wave_n0 = blockIdx.x * GS_BLOCK + wave * GS_WAVE;
        // prefilled outputs: rows of lanes that saw no camera hold zeros already.  The staged block stores below skip every
        // 16-byte piece whose rows are all such rows (what a piece of a live row's neighbour carries along is its exact zeros)
        const bool keep = any_on || !view.prefilled;
        const unsigned long long live = view.prefilled ? __ballot(any_on) : ~0ull;
        auto piece_live = [&](uint32_t j, uint32_t row_floats) { // float4 j of a block of 64 rows of row_floats floats
            const uint32_t r0 = (4u * j) / row_floats, r1 = (4u * j + 3u) / row_floats;
            return (((live >> r0) | (live >> (r1 < 64u ? r1 : 63u))) & 1ull) != 0ull;
        };
        if (split) {
            // two gradient tensors: v_coeffs [N,1,3] and v_coeffs_rest [N,K-1,3].  Same idea: a full wave stages its 64 rows
            // in LDS (row stride 3 / RL - 3 floats: odd, conflict-free) and writes each tensor's 64 rows as one contiguous block
            constexpr int R1 = RL - 3;
            const uint32_t rest_len = row_len - 3u;
            if (CAN_T && R1 > 0 && rest_len == (uint32_t)R1 && wave_n0 + GS_WAVE <= N) { // wave-uniform
                float *w = reinterpret_cast<float *>(s_tr + wave * GS_WAVE * (RL / 4));
#pragma unroll
                for (int i = 0; i < 3; ++i) w[lane * 3 + i] = acc[i];
#pragma unroll
                for (int i = 0; i < R1; ++i) w[GS_WAVE * 3 + lane * R1 + i] = acc[3 + i];
                __builtin_amdgcn_wave_barrier();
                const float4 *w4 = reinterpret_cast<const float4 *>(w);
                float4 *d0 = reinterpret_cast<float4 *>(v_coeffs + (size_t)wave_n0 * 3);
                if (lane < GS_WAVE * 3 / 4 && piece_live(lane, 3u)) d0[lane] = w4[lane];
                float4 *d1 = reinterpret_cast<float4 *>(view.v_coeffs_rest + (size_t)wave_n0 * R1);
                constexpr int N4 = GS_WAVE * R1 / 4; // (64 * R1 is a multiple of 4)
#pragma unroll
                for (int i = 0; i < (N4 + GS_WAVE - 1) / GS_WAVE; ++i)
                    if (i * GS_WAVE + (int)lane < N4 && piece_live(i * GS_WAVE + lane, (uint32_t)R1))
                        d1[i * GS_WAVE + lane] = w4[GS_WAVE * 3 / 4 + i * GS_WAVE + lane];
            } else if (keep) {
                float *o0 = v_coeffs + 3 * (size_t)n;
                o0[0] = acc[0]; o0[1] = acc[1]; o0[2] = acc[2];
                float *o1 = view.v_coeffs_rest + (size_t)n * rest_len;
#pragma unroll
                for (int i = 0; i < R1; ++i) o1[i] = acc[3 + i];
                for (uint32_t i = R1; i < rest_len; ++i) o1[i] = 0.f;
            }
        } else
        if (CAN_T && row_len == (uint32_t)RL && wave_n0 + GS_WAVE <= N) { // wave-uniform
            constexpr int NV = RL / 4;
            float4 *w = s_tr + wave * GS_WAVE * NV;
#pragma unroll
            for (int i = 0; i < NV; ++i) w[lane * NV + i] = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
            __builtin_amdgcn_wave_barrier();
            float4 *dst = reinterpret_cast<float4 *>(v_coeffs + (size_t)wave_n0 * RL);
#pragma unroll
            for (int i = 0; i < NV; ++i)
                if (piece_live(i * GS_WAVE + lane, (uint32_t)RL)) {
                    dst[i * GS_WAVE + lane] = w[i * GS_WAVE + lane];
                }
        } else if (keep) {
            store_row<NB * 3, VEC>(v_coeffs + (size_t)n * row_len, acc, row_len);
        }
    }
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
