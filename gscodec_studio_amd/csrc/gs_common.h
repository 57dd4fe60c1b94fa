// gs_common.h -- shared host/device helpers for libgsplat_hip (gfx950 only).
//
// Conventions (differ from the reference, which uses column-major glm types):
//   * 3x3 matrices are row-major structs (m[r][c]); symmetric 3x3 are 6 scalars
//     (xx, xy, xz, yy, yz, zz); symmetric 2x2 are 3 scalars (xx, xy, yy).
//   * quaternions are (w, x, y, z), not necessarily normalised.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gsplat_hip.h"

#define GS_WAVE 64
#define GS_BLOCK 256

#define GS_DEV __device__ __forceinline__
// First statement of a device function whose results must not depend on the kernel it is inlined into: no fma contraction of its
// operations.  The default, -ffp-contract=fast-honor-pragmas, fuses a multiply into an add wherever the surrounding code lets it, so
// the same source can round differently in two kernels -- and the __fmul_rn / __fadd_rn wrappers of this toolchain are plain
// operators (__clang_hip_math.h) that do NOT prevent it.  Carried by the arithmetic that has an exact counterpart elsewhere: the
// quantizers (quant_dev.h: bit-exact against torch's op-by-op arithmetic), the temporal slice (dynamic_dev.h: the stand-alone slice
// kernels and the projection kernels that evaluate it in their load phase) and tile_box below.  The projection chain itself stays
// contractible on purpose: with fused multiply-adds its gradients are measurably closer to float64 (round 6 tried it contraction-free:
// the quaternion gradient of tests/test_gpu_ops.py::test_projection_vs_golden_and_oracle went from 3.7e-5 to 1.0e-4 of float64,
// written with explicit fmas its worst entry was still 5x further off than the contracted code's).
#define GS_FP_STRICT _Pragma("clang fp contract(off)")

// ---------------------------------------------------------------------------
// error plumbing
// ---------------------------------------------------------------------------
void gs_set_error(const char *fmt, ...);

#define GS_CHECK_ARG(cond, msg)                                                        \
    do {                                                                               \
        if (!(cond)) {                                                                 \
            gs_set_error("%s: %s", __func__, msg);                                     \
            return 1;                                                                  \
        }                                                                              \
    } while (0)

#define GS_CHECK_LAUNCH()                                                              \
    do {                                                                               \
        hipError_t e_ = hipGetLastError();                                             \
        if (e_ != hipSuccess) {                                                        \
            gs_set_error("%s: launch failed: %s", __func__, hipGetErrorString(e_));    \
            return 2;                                                                  \
        }                                                                              \
    } while (0)

static inline uint32_t gs_div_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }

// Bucketed depth pre-sort (radix_sort.hip, isect.hip): bucket of a 64-bit (depth bits << 32 | element) key = the number of
// splitters <= key, over a table of 255 ascending splitters padded with UINT64_MAX (slot 255 is never read): 8 LDS reads.
#define GS_PRESORT_BUCKETS 256
#ifdef __HIPCC__
__device__ __forceinline__ uint32_t gs_bucket_of(const uint64_t *s_split, uint64_t key) {
    uint32_t lo = 0;
#pragma unroll
    for (uint32_t step = GS_PRESORT_BUCKETS / 2; step >= 1; step >>= 1)
        if (s_split[lo + step - 1] <= key) lo += step;
    return lo;
}
#endif

// The shN mask of the compression-simulation hooks (ada_mask.hip; fused into the SH evaluation by projection.hip / sh.hip):
// sigmoid(logit / T) in training mode, sigmoid(logit) >= 0.5 in eval mode -- IEEE operations in torch's order.
#ifdef __HIPCC__
__device__ __forceinline__ float gs_mask_sigmoid(float v) { return __fdiv_rn(1.f, __fadd_rn(1.f, expf(-v))); }
__device__ __forceinline__ float gs_mask_value(float logit, float temperature, int binary) {
    if (binary) return gs_mask_sigmoid(logit) >= 0.5f ? 1.f : 0.f;
    return gs_mask_sigmoid(__fdiv_rn(logit, temperature));
}
#endif

// Tile rectangle of a projected splat (isect_tiles.cu:56-69; the reference casts a possibly negative float to uint32 and
// relies on the saturating conversion, here the clamp is explicit).  No fma contraction inside (pragma; this toolchain's __f*_rn wrappers are
// plain operators and would be fused): the same values in every translation unit, whatever its contraction setting (isect.hip counts and emits with it, projection.hip counts with it).
#ifdef __HIPCC__
struct TileBox {
    int32_t x0, y0, x1, y1; // min inclusive, max exclusive
};
__device__ __forceinline__ TileBox tile_box(float mx, float my, int32_t radius, float tile_size, int32_t tw, int32_t th) {
    _Pragma("clang fp contract(off)");
    const float tr = ((float)radius / tile_size);
    const float tx = (mx / tile_size);
    const float ty = (my / tile_size);
    TileBox b;
    b.x0 = min(max(0, (int32_t)floorf((tx - tr))), tw);
    b.y0 = min(max(0, (int32_t)floorf((ty - tr))), th);
    b.x1 = min(max(0, (int32_t)ceilf((tx + tr))), tw);
    b.y1 = min(max(0, (int32_t)ceilf((ty + tr))), th);
    return b;
}
#endif

// radix_sort.hip: slot of the first pass's [256][n_blocks] digit histogram inside a sort's temp buffer (nullptr: not applicable)
uint32_t *sort_first_hist_slot(uint64_t n, void *temp, size_t temp_bytes, uint32_t *n_blocks);

// ---------------------------------------------------------------------------
// small linear algebra
// ---------------------------------------------------------------------------
struct Vec3 {
    float x, y, z;
};
struct Mat3 {
    float m[3][3];
};
struct Sym3 {
    float xx, xy, xz, yy, yz, zz;
};
struct Sym2 {
    float xx, xy, yy;
};

GS_DEV Mat3 mat3_zero() {
    Mat3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.m[i][j] = 0.f;
    return r;
}

GS_DEV Mat3 sym3_to_mat3(const Sym3 &s) {
    Mat3 r;
    r.m[0][0] = s.xx; r.m[0][1] = s.xy; r.m[0][2] = s.xz;
    r.m[1][0] = s.xy; r.m[1][1] = s.yy; r.m[1][2] = s.yz;
    r.m[2][0] = s.xz; r.m[2][1] = s.yz; r.m[2][2] = s.zz;
    return r;
}

GS_DEV Mat3 mat3_mul(const Mat3 &a, const Mat3 &b) {
    Mat3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}

// a^T * b
GS_DEV Mat3 mat3_tmul(const Mat3 &a, const Mat3 &b) {
    Mat3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r.m[i][j] = a.m[0][i] * b.m[0][j] + a.m[1][i] * b.m[1][j] + a.m[2][i] * b.m[2][j];
    return r;
}

// a * b^T
GS_DEV Mat3 mat3_mult(const Mat3 &a, const Mat3 &b) {
    Mat3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            r.m[i][j] = a.m[i][0] * b.m[j][0] + a.m[i][1] * b.m[j][1] + a.m[i][2] * b.m[j][2];
    return r;
}

// W * S * W^T for symmetric S -> symmetric
GS_DEV Sym3 sym3_congruence(const Mat3 &W, const Sym3 &S) {
    Mat3 Sm = sym3_to_mat3(S);
    Mat3 WS = mat3_mul(W, Sm);
    Sym3 r;
    r.xx = WS.m[0][0] * W.m[0][0] + WS.m[0][1] * W.m[0][1] + WS.m[0][2] * W.m[0][2];
    r.xy = WS.m[0][0] * W.m[1][0] + WS.m[0][1] * W.m[1][1] + WS.m[0][2] * W.m[1][2];
    r.xz = WS.m[0][0] * W.m[2][0] + WS.m[0][1] * W.m[2][1] + WS.m[0][2] * W.m[2][2];
    r.yy = WS.m[1][0] * W.m[1][0] + WS.m[1][1] * W.m[1][1] + WS.m[1][2] * W.m[1][2];
    r.yz = WS.m[1][0] * W.m[2][0] + WS.m[1][1] * W.m[2][1] + WS.m[1][2] * W.m[2][2];
    r.zz = WS.m[2][0] * W.m[2][0] + WS.m[2][1] * W.m[2][1] + WS.m[2][2] * W.m[2][2];
    return r;
}

// W^T * G * W for symmetric G -> symmetric
GS_DEV Sym3 sym3_congruence_t(const Mat3 &W, const Sym3 &G) {
    Mat3 Gm = sym3_to_mat3(G);
    Mat3 GW = mat3_mul(Gm, W); // G * W
    Sym3 r;
    r.xx = W.m[0][0] * GW.m[0][0] + W.m[1][0] * GW.m[1][0] + W.m[2][0] * GW.m[2][0];
    r.xy = W.m[0][0] * GW.m[0][1] + W.m[1][0] * GW.m[1][1] + W.m[2][0] * GW.m[2][1];
    r.xz = W.m[0][0] * GW.m[0][2] + W.m[1][0] * GW.m[1][2] + W.m[2][0] * GW.m[2][2];
    r.yy = W.m[0][1] * GW.m[0][1] + W.m[1][1] * GW.m[1][1] + W.m[2][1] * GW.m[2][1];
    r.yz = W.m[0][1] * GW.m[0][2] + W.m[1][1] * GW.m[1][2] + W.m[2][1] * GW.m[2][2];
    r.zz = W.m[0][2] * GW.m[0][2] + W.m[1][2] * GW.m[1][2] + W.m[2][2] * GW.m[2][2];
    return r;
}

// rotation matrix of a (possibly un-normalised) quaternion (w,x,y,z).
// reference behaviour: gsplat/cuda/include/quat.cuh:9-31
GS_DEV Mat3 quat_to_rotmat(float w, float x, float y, float z) {
    float inv = rsqrtf(w * w + x * x + y * y + z * z);
    w *= inv; x *= inv; y *= inv; z *= inv;
    float xx = x * x, yy = y * y, zz = z * z;
    float xy = x * y, xz = x * z, yz = y * z;
    float wx = w * x, wy = w * y, wz = w * z;
    Mat3 R;
    R.m[0][0] = 1.f - 2.f * (yy + zz); R.m[0][1] = 2.f * (xy - wz);       R.m[0][2] = 2.f * (xz + wy);
    R.m[1][0] = 2.f * (xy + wz);       R.m[1][1] = 1.f - 2.f * (xx + zz); R.m[1][2] = 2.f * (yz - wx);
    R.m[2][0] = 2.f * (xz - wy);       R.m[2][1] = 2.f * (yz + wx);       R.m[2][2] = 1.f - 2.f * (xx + yy);
    return R;
}

// Sigma = (R S)(R S)^T   (gsplat/cuda/include/quat_scale_to_covar_preci.cuh:10-41)
GS_DEV Sym3 covar_from_rot_scale(const Mat3 &R, float sx, float sy, float sz) {
    Mat3 M;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        M.m[i][0] = R.m[i][0] * sx;
        M.m[i][1] = R.m[i][1] * sy;
        M.m[i][2] = R.m[i][2] * sz;
    }
    Sym3 S;
    S.xx = M.m[0][0] * M.m[0][0] + M.m[0][1] * M.m[0][1] + M.m[0][2] * M.m[0][2];
    S.xy = M.m[0][0] * M.m[1][0] + M.m[0][1] * M.m[1][1] + M.m[0][2] * M.m[1][2];
    S.xz = M.m[0][0] * M.m[2][0] + M.m[0][1] * M.m[2][1] + M.m[0][2] * M.m[2][2];
    S.yy = M.m[1][0] * M.m[1][0] + M.m[1][1] * M.m[1][1] + M.m[1][2] * M.m[1][2];
    S.yz = M.m[1][0] * M.m[2][0] + M.m[1][1] * M.m[2][1] + M.m[1][2] * M.m[2][2];
    S.zz = M.m[2][0] * M.m[2][0] + M.m[2][1] * M.m[2][1] + M.m[2][2] * M.m[2][2];
    return S;
}

// VJP of Sigma = (R S)(R S)^T w.r.t. quaternion and scale, given a SYMMETRISED
// upstream gradient G (G = v_Sigma + v_Sigma^T already folded: pass the full
// matrix gradient as Mat3, not assumed symmetric).
// reference behaviour: quat_scale_to_covar_preci.cuh:43-81, quat.cuh:33-57
GS_DEV void covar_vjp_quat_scale(
    float qw, float qx, float qy, float qz, float sx, float sy, float sz,
    const Mat3 &R, const Mat3 &vSigma, float vq[4], float vs[3]) {
    // M = R S ; v_M = (G + G^T) M
    Mat3 Gs;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Gs.m[i][j] = vSigma.m[i][j] + vSigma.m[j][i];
    Mat3 M;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        M.m[i][0] = R.m[i][0] * sx;
        M.m[i][1] = R.m[i][1] * sy;
        M.m[i][2] = R.m[i][2] * sz;
    }
    Mat3 vM = mat3_mul(Gs, M);
    // v_s_j = sum_i R_ij vM_ij ; v_R_ij = vM_ij * s_j
    vs[0] += R.m[0][0] * vM.m[0][0] + R.m[1][0] * vM.m[1][0] + R.m[2][0] * vM.m[2][0];
    vs[1] += R.m[0][1] * vM.m[0][1] + R.m[1][1] * vM.m[1][1] + R.m[2][1] * vM.m[2][1];
    vs[2] += R.m[0][2] * vM.m[0][2] + R.m[1][2] * vM.m[1][2] + R.m[2][2] * vM.m[2][2];
    Mat3 V;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        V.m[i][0] = vM.m[i][0] * sx;
        V.m[i][1] = vM.m[i][1] * sy;
        V.m[i][2] = vM.m[i][2] * sz;
    }
    // d R / d q_normalised
    float inv = rsqrtf(qw * qw + qx * qx + qy * qy + qz * qz);
    float w = qw * inv, x = qx * inv, y = qy * inv, z = qz * inv;
    float gw = 2.f * (x * (V.m[2][1] - V.m[1][2]) + y * (V.m[0][2] - V.m[2][0]) + z * (V.m[1][0] - V.m[0][1]));
    float gx = 2.f * (-2.f * x * (V.m[1][1] + V.m[2][2]) + y * (V.m[1][0] + V.m[0][1]) +
                      z * (V.m[2][0] + V.m[0][2]) + w * (V.m[2][1] - V.m[1][2]));
    float gy = 2.f * (x * (V.m[1][0] + V.m[0][1]) - 2.f * y * (V.m[0][0] + V.m[2][2]) +
                      z * (V.m[2][1] + V.m[1][2]) + w * (V.m[0][2] - V.m[2][0]));
    float gz = 2.f * (x * (V.m[2][0] + V.m[0][2]) + y * (V.m[2][1] + V.m[1][2]) -
                      2.f * z * (V.m[0][0] + V.m[1][1]) + w * (V.m[1][0] - V.m[0][1]));
    // through the normalisation: (g - (g . qn) qn) / |q|
    float dot = gw * w + gx * x + gy * y + gz * z;
    vq[0] += (gw - dot * w) * inv;
    vq[1] += (gx - dot * x) * inv;
    vq[2] += (gy - dot * y) * inv;
    vq[3] += (gz - dot * z) * inv;
}

// ---------------------------------------------------------------------------
// wave64 helpers
// ---------------------------------------------------------------------------
GS_DEV uint32_t lane_id() { return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

// full-wave sum, result valid in every lane
GS_DEV float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

GS_DEV int wave_max_i32(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        int o = __shfl_xor(v, off, 64);
        v = v > o ? v : o;
    }
    return v;
}
