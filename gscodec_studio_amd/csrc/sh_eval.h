// sh_eval.h -- the pieces of the spherical-harmonics evaluation shared by sh.hip (the stand-alone kernels) and
// projection.hip (the row projection evaluates the colour of a splat in the same pass).  Device code only.
#pragma once

#include "gs_common.h"

namespace {

template <int DEG>
struct ShDim {
    static constexpr int NB = (DEG + 1) * (DEG + 1);
};

// Real SH basis values for a UNIT direction (x,y,z); Sloan's fast evaluation constants
// (spherical_harmonics.cuh:22-96).
template <int DEG>
GS_DEV void sh_basis(float x, float y, float z, float *Y) {
    Y[0] = 0.2820947917738781f;
    if (DEG < 1) return;
    Y[1] = -0.48860251190292f * y;
    Y[2] = 0.48860251190292f * z;
    Y[3] = -0.48860251190292f * x;
    if (DEG < 2) return;
    float z2 = z * z;
    float c1 = x * x - y * y, s1 = 2.f * x * y;
    float t0b = -1.092548430592079f * z;
    Y[4] = 0.5462742152960395f * s1;
    Y[5] = t0b * y;
    Y[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
    Y[7] = t0b * x;
    Y[8] = 0.5462742152960395f * c1;
    if (DEG < 3) return;
    float c2 = x * c1 - y * s1, s2 = x * s1 + y * c1;
    float t0c = -2.285228997322329f * z2 + 0.4570457994644658f;
    float t1b = 1.445305721320277f * z;
    Y[9] = -0.5900435899266435f * s2;
    Y[10] = t1b * s1;
    Y[11] = t0c * y;
    Y[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
    Y[13] = t0c * x;
    Y[14] = t1b * c1;
    Y[15] = -0.5900435899266435f * c2;
    if (DEG < 4) return;
    float c3 = x * c2 - y * s2, s3 = x * s2 + y * c2;
    float t0d = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    float t1c = 3.31161143515146f * z2 - 0.47308734787878f;
    float t2b = -1.770130769779931f * z;
    Y[16] = 0.6258357354491763f * s3;
    Y[17] = t2b * s2;
    Y[18] = t1c * s1;
    Y[19] = t0d * y;
    Y[20] = 1.984313483298443f * z * Y[12] - 1.006230589874905f * Y[6];
    Y[21] = t0d * x;
    Y[22] = t1c * c1;
    Y[23] = t2b * c2;
    Y[24] = 0.6258357354491763f * c3;
}

// row I/O: VEC => the row base is 16-byte aligned and its length (3K floats) is a
// multiple of 4, so dwordx4 accesses are legal for every row.
template <int CNT, bool VEC>
GS_DEV void load_floats(const float *__restrict__ p, float *dst) {
    if (VEC) {
        constexpr int NV = CNT / 4;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            float4 v = reinterpret_cast<const float4 *>(p)[i];
            dst[4 * i] = v.x; dst[4 * i + 1] = v.y; dst[4 * i + 2] = v.z; dst[4 * i + 3] = v.w;
        }
#pragma unroll
        for (int i = NV * 4; i < CNT; ++i) dst[i] = p[i];
    } else {
#pragma unroll
        for (int i = 0; i < CNT; ++i) dst[i] = p[i];
    }
}

// -A^-1 t of the affine world->camera matrix V = [[A, t], [0, 1]] (adjugate form): the camera centre in world space
GS_DEV void camera_center(const float *__restrict__ V, float &x, float &y, float &z) {
    float a00 = V[0], a01 = V[1], a02 = V[2], t0 = V[3];
    float a10 = V[4], a11 = V[5], a12 = V[6], t1 = V[7];
    float a20 = V[8], a21 = V[9], a22 = V[10], t2 = V[11];
    // rows of adj(A) = cross products of the columns of A
    float r00 = a11 * a22 - a21 * a12, r01 = a21 * a02 - a01 * a22, r02 = a01 * a12 - a11 * a02;
    float r10 = a12 * a20 - a22 * a10, r11 = a22 * a00 - a02 * a20, r12 = a02 * a10 - a12 * a00;
    float r20 = a10 * a21 - a20 * a11, r21 = a20 * a01 - a00 * a21, r22 = a00 * a11 - a10 * a01;
    float inv = 1.f / (a00 * r00 + a10 * r01 + a20 * r02);
    x = -(r00 * t0 + r01 * t1 + r02 * t2) * inv;
    y = -(r10 * t0 + r11 * t1 + r12 * t2) * inv;
    z = -(r20 * t0 + r21 * t1 + r22 * t2) * inv;
}

// SPLIT coefficient rows: the trainers keep the DC band and the higher bands as two parameters, sh0 [N,1,3] and shN [N,K-1,3]
// (reference examples/simple_trainer.py:779-786 concatenates them before every render: 193 MB each way at 1 M splats, and
// autograd splits the gradient again).  The kernels take the two tensors as they are: the first 3 floats of a row come from
// `row`, the rest from `rest` (rows of 3 (K-1) floats: only 4-byte aligned, read with dword-aligned 16-byte loads).
struct __attribute__((packed, aligned(4))) float4_a4 {
    float x, y, z, w;
};

template <int CNT>
GS_DEV void load_coeff_row_split(const float *__restrict__ row, const float *__restrict__ rest, float *dst) {
    dst[0] = row[0];
    if (CNT > 1) dst[CNT > 1 ? 1 : 0] = row[1];
    if (CNT > 2) dst[CNT > 2 ? 2 : 0] = row[2];
    constexpr int R = CNT > 3 ? CNT - 3 : 0;
    constexpr int NV = R / 4;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        const float4_a4 v = reinterpret_cast<const float4_a4 *>(rest)[i];
        dst[3 + 4 * i] = v.x; dst[3 + 4 * i + 1] = v.y; dst[3 + 4 * i + 2] = v.z; dst[3 + 4 * i + 3] = v.w;
    }
#pragma unroll
    for (int i = NV * 4; i < R; ++i) dst[3 + i] = rest[i];
}

template <int CNT, bool VEC>
GS_DEV void load_coeff_row(const float *__restrict__ row, const float *__restrict__ rest, float *dst) {
    if (rest == nullptr) load_floats<CNT, VEC>(row, dst); // (uniform) one contiguous row
    else load_coeff_row_split<CNT>(row, rest, dst);
}

// clamp_min(SH colour + 0.5, 0) of ONE splat seen from direction (dx, dy, dz) (not normalised), coefficient row `row`
// ([K,3], the first (DEG+1)^2 bands are used): the arithmetic of sh_fwd_kernel's view mode, shared so that the projection's
// fused colour is bit-identical to gs_sh_view_fwd's.
// LAYOUT: 0 = one row (VEC: 16-byte aligned), 1 = split rows, -1 = decided at run time by `rest`
// band_mask (optional): the coefficients of the bands >= 1 are multiplied by *band_mask first (the shN mask of the
// compression-simulation hooks, one rounding per coefficient: what the stand-alone mask kernel would have stored).
template <int DEG, bool VEC, int LAYOUT = -1>
GS_DEV void sh_view_color(float dx, float dy, float dz, const float *__restrict__ row, const float *__restrict__ rest, bool clamp_half,
                          float &r, float &g, float &b, const float *band_mask = nullptr) {
    constexpr int NB = ShDim<DEG>::NB;
    float Y[NB];
    if (DEG >= 1) {
        float inv = rsqrtf(dx * dx + dy * dy + dz * dz);
        sh_basis<DEG>(dx * inv, dy * inv, dz * inv, Y);
    } else {
        sh_basis<0>(0.f, 0.f, 1.f, Y);
    }
    float cf[NB * 3];
    if (LAYOUT == 0) load_floats<NB * 3, VEC>(row, cf);
    else if (LAYOUT == 1) load_coeff_row_split<NB * 3>(row, rest, cf);
    else load_coeff_row<NB * 3, VEC>(row, rest, cf);
    if (band_mask != nullptr) {
        const float m = *band_mask;
#pragma unroll
        for (int i = 3; i < NB * 3; ++i) cf[i] = __fmul_rn(cf[i], m);
    }
    r = g = b = 0.f;
#pragma unroll
    for (int k = 0; k < NB; ++k) {
        r += Y[k] * cf[3 * k];
        g += Y[k] * cf[3 * k + 1];
        b += Y[k] * cf[3 * k + 2];
    }
    if (clamp_half) {
        r = fmaxf(r + 0.5f, 0.f);
        g = fmaxf(g + 0.5f, 0.f);
        b = fmaxf(b + 0.5f, 0.f);
    }
}

} // namespace
