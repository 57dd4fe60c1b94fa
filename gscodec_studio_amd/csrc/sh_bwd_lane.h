// sh_bwd_lane.h -- the per-gaussian SH backward shared by sh.hip (its own kernel) and projection.hip (fused behind the
// projection backward, round 3).  See sh.hip for the design notes.
#pragma once
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
    // the shN mask applied by the forward (split rows): effective coefficients of the bands >= 1 = raw * mask.  bwd: the
    // gradient of the raw coefficients is v_effective * mask, the logit's (v_effective . raw) * mask (1 - mask) / T, reduced
    // in the lane (over cameras and coefficients) in the order of the stand-alone mask kernel
    const float *mask_logits; // [N] or NULL
    float mask_temp;
    int mask_binary;
    float *v_mask_logits;     // [N] (every entry written) or NULL
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

// The SH backward of ONE gaussian (lane): loops over cameras.  SHARED: v_coeffs is [N,K,3] and the lane accumulates over
// cameras in registers; otherwise [C,N,K,3] rows are written per camera.  Used by sh_bwd_kernel (sh.hip) and, fused behind
// the projection backward, by projection_bwd_kernel (projection.hip).  Lanes with in_range == false take part in the wave's
// ballots and staged stores but own no gaussian.  Returns d/d means (sum over cameras of d/d dirs) and whether any camera
// saw the gaussian.
template <int DEG, bool VEC, bool SHARED>
GS_DEV void sh_bwd_lane(
    uint32_t C, uint32_t N, uint32_t K, uint32_t n, bool in_range, const float *__restrict__ dirs,
    const float *__restrict__ coeffs, const uint8_t *__restrict__ masks,
    const float *__restrict__ v_colors, float *__restrict__ v_coeffs,
    float *__restrict__ v_dirs, const ShView &view, const float *__restrict__ colors_out, uint32_t v_colors_stride,
    bool want_dir_grad /* d/d dirs is wanted (v_dirs, or its sum over cameras through vmx..vmz) */,
    float &vmx, float &vmy, float &vmz, bool &any_on) {
    constexpr int NB = ShDim<DEG>::NB;
    vmx = vmy = vmz = 0.f; // view mode: d/d means = sum over cameras of d/d dirs
    if (in_range && view.v_opac_out != nullptr) { // culled (c, n) pairs hold exact zeros in the gradient rows
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
    any_on = false;
    const bool masked = SHARED && split && view.mask_logits != nullptr; // (uniform)
    const float band_mask = (masked && in_range) ? gs_mask_value(view.mask_logits[n], view.mask_temp, view.mask_binary) : 1.f;
    for (uint32_t c = 0; c < C; ++c) {
        size_t e = (size_t)c * N + n;
        bool on = in_range && sh_active(masks, view, e);
        any_on |= on;
        if (!on) {
            if (!in_range) continue;
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
        if (want_dir_grad) {
            float gx = 0.f, gy = 0.f, gz = 0.f;
            if (DEG >= 1) {
                if (!SHARED || !have_cf) {
                    if (split) load_coeff_row<NB * 3, VEC>(coeffs + 3 * (size_t)n, view.coeffs_rest + (size_t)n * (K - 1) * 3, cf);
                    else load_floats<NB * 3, VEC>(coeffs + (SHARED ? (size_t)n : e) * row_len, cf);
                    have_cf = true;
                }
                float w[NB];
                if (masked) { // (cf holds the RAW coefficients: the effective ones are rounded products, as the forward used them)
                    w[0] = cf[0] * vr + cf[1] * vg + cf[2] * vb;
#pragma unroll
                    for (int k = 1; k < NB; ++k)
                        w[k] = __fmul_rn(cf[3 * k], band_mask) * vr + __fmul_rn(cf[3 * k + 1], band_mask) * vg + __fmul_rn(cf[3 * k + 2], band_mask) * vb;
                } else {
#pragma unroll
                for (int k = 0; k < NB; ++k) w[k] = cf[3 * k] * vr + cf[3 * k + 1] * vg + cf[3 * k + 2] * vb;
                }
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
    if (masked) {
        // acc holds d/d (effective coefficients), summed over the cameras
        if (view.v_mask_logits != nullptr && in_range) {
            float v_logit = 0.f;
            if (any_on) {
                if (!have_cf) load_coeff_row<NB * 3, VEC>(coeffs + 3 * (size_t)n, view.coeffs_rest + (size_t)n * (K - 1) * 3, cf);
                float dot = 0.f;
#pragma unroll
                for (int i = 3; i < NB * 3; ++i) dot = __fadd_rn(dot, __fmul_rn(acc[i], cf[i]));
                v_logit = __fdiv_rn(__fmul_rn(__fmul_rn(dot, __fsub_rn(1.f, band_mask)), band_mask), view.mask_temp);
            }
            view.v_mask_logits[n] = v_logit;
        }
#pragma unroll
        for (int i = 3; i < NB * 3; ++i) acc[i] = __fmul_rn(acc[i], band_mask);
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
        const bool keep = in_range && (any_on || !view.prefilled);
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
}

}  // namespace
