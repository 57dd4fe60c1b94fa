// projection_dev.h -- the per-(camera, gaussian) projection arithmetic and its VJP, shared by the kernels of projection.hip and
// projection_dyn.hip (the temporal slice of dynamic splats evaluated in the projection's load phase): ONE definition, so that
// every kernel that projects a splat computes bit-identical rows.  Device code only.
//   reference: gsplat/cuda/csrc/fully_fused_projection_fwd.cu:22-196, fully_fused_projection_bwd.cu:24-263
#pragma once
#include "gs_common.h"
#include "proj_models.h"

namespace {

GS_DEV Sym3 load_covar(
    const float *__restrict__ covars, const float *__restrict__ quats,
    const float *__restrict__ scales, uint32_t n) {
    if (covars != nullptr) {
        const float *c = covars + 6 * (size_t)n;
        Sym3 S = {c[0], c[1], c[2], c[3], c[4], c[5]};
        return S;
    }
    const float *q = quats + 4 * (size_t)n;
    const float *s = scales + 3 * (size_t)n;
    Mat3 R = quat_to_rotmat(q[0], q[1], q[2], q[3]);
    return covar_from_rot_scale(R, s[0], s[1], s[2]);
}

// cov2d = J Sigma J^T
GS_DEV Sym2 project_covar(const Jac &J, const Sym3 &S) {
    float a0 = J.j00 * S.xx + J.j01 * S.xy + J.j02 * S.xz;
    float a1 = J.j00 * S.xy + J.j01 * S.yy + J.j02 * S.yz;
    float a2 = J.j00 * S.xz + J.j01 * S.yz + J.j02 * S.zz;
    float b0 = J.j10 * S.xx + J.j11 * S.xy + J.j12 * S.xz;
    float b1 = J.j10 * S.xy + J.j11 * S.yy + J.j12 * S.yz;
    float b2 = J.j10 * S.xz + J.j11 * S.yz + J.j12 * S.zz;
    Sym2 c;
    c.xx = a0 * J.j00 + a1 * J.j01 + a2 * J.j02;
    c.xy = a0 * J.j10 + a1 * J.j11 + a2 * J.j12;
    c.yy = b0 * J.j10 + b1 * J.j11 + b2 * J.j12;
    return c;
}

struct Splat2D {
    int32_t radius; // 0 => culled
    float mx, my, depth, ca, cb, cc, comp;
};

// Shared forward maths.  PACKED selects the packed-kernel radius formula
// (fully_fused_projection_packed_fwd.cu:183-186) instead of the unpacked one
// (fully_fused_projection_fwd.cu:167-169).
// `covar()` is only evaluated for splats inside the depth range (its loads are skipped for the rest).
template <bool PACKED, class CovarFn>
GS_DEV Splat2D project_point(
    const Camera &cam, float px, float py, float pz, CovarFn covar,
    int W, int H, float eps2d, float near_plane, float far_plane, float radius_clip,
    int camera_model) {
    Splat2D out;
    out.radius = 0;
    float x = cam.W.m[0][0] * px + cam.W.m[0][1] * py + cam.W.m[0][2] * pz + cam.tx;
    float y = cam.W.m[1][0] * px + cam.W.m[1][1] * py + cam.W.m[1][2] * pz + cam.ty;
    float z = cam.W.m[2][0] * px + cam.W.m[2][1] * py + cam.W.m[2][2] * pz + cam.tz;
    if (z < near_plane || z > far_plane) return out;

    Sym3 S = covar();
    Sym3 Sc = sym3_congruence(cam.W, S);

    Jac J;
    float mx, my;
    if (camera_model == GS_CAMERA_PINHOLE) {
        float tx_, ty_;
        pinhole_jac(cam, x, y, z, W, H, J, mx, my, tx_, ty_);
    } else if (camera_model == GS_CAMERA_ORTHO) {
        ortho_jac(cam, x, y, J, mx, my);
    } else {
        fisheye_jac(cam, x, y, z, J, mx, my);
    }
    Sym2 c2 = project_covar(J, Sc);

    // blur + compensation (gsplat/cuda/include/utils.cuh:30-37)
    float det_orig = c2.xx * c2.yy - c2.xy * c2.xy;
    c2.xx += eps2d;
    c2.yy += eps2d;
    float det = c2.xx * c2.yy - c2.xy * c2.xy;
    if (det <= 0.f) return out;
    out.comp = sqrtf(fmaxf(0.f, det_orig / det));

    float inv_det = 1.f / det;
    out.ca = c2.yy * inv_det;
    out.cb = -c2.xy * inv_det;
    out.cc = c2.xx * inv_det;

    float b = 0.5f * (c2.xx + c2.yy);
    float radius;
    if (PACKED) {
        float sq = sqrtf(fmaxf(0.1f, b * b - det));
        float v1 = b + sq, v2 = b - sq;
        radius = ceilf(3.f * sqrtf(fmaxf(v1, v2)));
    } else {
        float v1 = b + sqrtf(fmaxf(0.01f, b * b - det));
        radius = ceilf(3.f * sqrtf(v1));
    }
    if (radius <= radius_clip) return out;
    if (mx + radius <= 0 || mx - radius >= W || my + radius <= 0 || my - radius >= H) return out;

    out.radius = (int32_t)radius;
    out.mx = mx; out.my = my; out.depth = z;
    return out;
}

template <bool PACKED>
GS_DEV Splat2D project_one(
    const Camera &cam, const float *__restrict__ means, const float *__restrict__ covars,
    const float *__restrict__ quats, const float *__restrict__ scales, uint32_t n,
    int W, int H, float eps2d, float near_plane, float far_plane, float radius_clip,
    int camera_model) {
    const float *p = means + 3 * (size_t)n;
    return project_point<PACKED>(cam, p[0], p[1], p[2], [&]() { return load_covar(covars, quats, scales, n); }, W, H, eps2d, near_plane,
                                 far_plane, radius_clip, camera_model);
}

// The binning's intersection count in the projection's own pass (what gs_isect_count_keys would compute from the rows afterwards):
// tiles_per_gauss[idx] and, per workgroup, (intersections, visible pairs) as ONE 8-byte store into block_sums (pinned host memory) --
// the host learns n_isects one kernel after the step starts.  EVERY thread of the workgroup must call this (block-wide sum).
GS_DEV void rows_count_tiles(const Splat2D &s, bool in, size_t idx, int32_t *__restrict__ tiles_per_gauss, int32_t *__restrict__ block_sums,
                             float tile_size, int32_t tile_width, int32_t tile_height) {
    int32_t cnt = 0;
    if (s.radius > 0) {
        const TileBox b = tile_box(s.mx, s.my, s.radius, tile_size, tile_width, tile_height);
        cnt = (b.y1 - b.y0) * (b.x1 - b.x0);
    }
    if (in) tiles_per_gauss[idx] = cnt;
    if (block_sums != nullptr) {
        __shared__ int32_t s_cnt[GS_BLOCK / GS_WAVE], s_vis[GS_BLOCK / GS_WAVE];
        int32_t v = cnt;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
        const int32_t nvis = (int32_t)__popcll(__ballot(s.radius > 0));
        if ((threadIdx.x & 63u) == 0u) {
            s_cnt[threadIdx.x >> 6] = v;
            s_vis[threadIdx.x >> 6] = nvis;
        }
        __syncthreads();
        if (threadIdx.x == 0) // (intersections, visible pairs) of the block as ONE 8-byte store: they reach the host together
            reinterpret_cast<int2 *>(block_sums)[blockIdx.y * gridDim.x + blockIdx.x] =
                make_int2(s_cnt[0] + s_cnt[1] + s_cnt[2] + s_cnt[3], s_vis[0] + s_vis[1] + s_vis[2] + s_vis[3]);
    }
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------
struct ProjGrad {
    float v_px, v_py, v_pz; // d/d mean (world)
    Sym3 v_S;               // d/d Sigma (world), symmetric full-matrix gradient
    Mat3 v_W;               // d/d viewmat rotation
    float v_t[3];           // d/d viewmat translation
};

// VJP for one (camera, gaussian) pair.  Follows the chain of
// fully_fused_projection_bwd.cu:68-199 on symmetric forms.
template <bool NEED_VIEW>
GS_DEV void project_one_vjp(
    const Camera &cam, float px, float py, float pz, const Sym3 &S,
    int W, int H, float eps2d, int camera_model,
    float ca, float cb, float cc,              // conic
    float comp, float v_comp, bool has_comp,   // compensation
    float v_mx, float v_my, float v_depth, float v_ca, float v_cb, float v_cc,
    ProjGrad &g) {
    // conic = inverse(cov2d_blur): v_cov = -P G P, G = [[v_ca, v_cb/2],[v_cb/2, v_cc]]
    float g01 = 0.5f * v_cb;
    float t00 = ca * v_ca + cb * g01, t01 = ca * g01 + cb * v_cc;
    float t10 = cb * v_ca + cc * g01, t11 = cb * g01 + cc * v_cc;
    Sym2 G;
    G.xx = -(t00 * ca + t01 * cb);
    G.xy = -(t00 * cb + t01 * cc);
    G.yy = -(t10 * cb + t11 * cc);
    if (has_comp) {
        // utils.cuh:39-73
        float det_conic = ca * cc - cb * cb;
        float v_sqr = v_comp * 0.5f / (comp + 1e-6f);
        float om = 1.f - comp * comp;
        G.xx += v_sqr * (om * ca - eps2d * det_conic);
        G.xy += v_sqr * (om * cb);
        G.yy += v_sqr * (om * cc - eps2d * det_conic);
    }

    float x = cam.W.m[0][0] * px + cam.W.m[0][1] * py + cam.W.m[0][2] * pz + cam.tx;
    float y = cam.W.m[1][0] * px + cam.W.m[1][1] * py + cam.W.m[1][2] * pz + cam.ty;
    float z = cam.W.m[2][0] * px + cam.W.m[2][1] * py + cam.W.m[2][2] * pz + cam.tz;
    Sym3 Sc = sym3_congruence(cam.W, S);

    Jac J;
    float mx, my, txc = 0.f, tyc = 0.f;
    FisheyeTerms ft;
    if (camera_model == GS_CAMERA_PINHOLE) {
        pinhole_jac(cam, x, y, z, W, H, J, mx, my, txc, tyc);
    } else if (camera_model == GS_CAMERA_ORTHO) {
        ortho_jac(cam, x, y, J, mx, my);
    } else {
        fisheye_jac(cam, x, y, z, J, mx, my);
        ft = fisheye_terms(x, y, z);
    }

    // v_Sc = J^T G J (symmetric 3x3)
    float gj00 = G.xx * J.j00 + G.xy * J.j10, gj01 = G.xx * J.j01 + G.xy * J.j11, gj02 = G.xx * J.j02 + G.xy * J.j12;
    float gj10 = G.xy * J.j00 + G.yy * J.j10, gj11 = G.xy * J.j01 + G.yy * J.j11, gj12 = G.xy * J.j02 + G.yy * J.j12;
    Sym3 vSc;
    vSc.xx = J.j00 * gj00 + J.j10 * gj10;
    vSc.xy = J.j00 * gj01 + J.j10 * gj11;
    vSc.xz = J.j00 * gj02 + J.j10 * gj12;
    vSc.yy = J.j01 * gj01 + J.j11 * gj11;
    vSc.yz = J.j01 * gj02 + J.j11 * gj12;
    vSc.zz = J.j02 * gj02 + J.j12 * gj12;

    // v_J = 2 G J Sc  (2x3)
    float vj00 = 2.f * (gj00 * Sc.xx + gj01 * Sc.xy + gj02 * Sc.xz);
    float vj01 = 2.f * (gj00 * Sc.xy + gj01 * Sc.yy + gj02 * Sc.yz);
    float vj02 = 2.f * (gj00 * Sc.xz + gj01 * Sc.yz + gj02 * Sc.zz);
    float vj10 = 2.f * (gj10 * Sc.xx + gj11 * Sc.xy + gj12 * Sc.xz);
    float vj11 = 2.f * (gj10 * Sc.xy + gj11 * Sc.yy + gj12 * Sc.yz);
    float vj12 = 2.f * (gj10 * Sc.xz + gj11 * Sc.yz + gj12 * Sc.zz);

    float vx, vy, vz; // d/d pc
    proj_mean_vjp(cam, camera_model, x, y, z, W, H, v_mx, v_my, vj00, vj01, vj02, vj10, vj11, vj12, vx, vy, vz);
    vz += v_depth;

    // world->camera VJP (gsplat/cuda/include/transform.cuh:19-69)
    g.v_px += cam.W.m[0][0] * vx + cam.W.m[1][0] * vy + cam.W.m[2][0] * vz;
    g.v_py += cam.W.m[0][1] * vx + cam.W.m[1][1] * vy + cam.W.m[2][1] * vz;
    g.v_pz += cam.W.m[0][2] * vx + cam.W.m[1][2] * vy + cam.W.m[2][2] * vz;
    Sym3 vS = sym3_congruence_t(cam.W, vSc);
    g.v_S.xx += vS.xx; g.v_S.xy += vS.xy; g.v_S.xz += vS.xz;
    g.v_S.yy += vS.yy; g.v_S.yz += vS.yz; g.v_S.zz += vS.zz;
    if (NEED_VIEW) {
        // v_W = v_pc p^T + 2 vSc W S
        Mat3 WS = mat3_mul(cam.W, sym3_to_mat3(S));
        Mat3 vScm = sym3_to_mat3(vSc);
        Mat3 t = mat3_mul(vScm, WS);
        float vp[3] = {vx, vy, vz};
        float pw[3] = {px, py, pz};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) g.v_W.m[i][j] += vp[i] * pw[j] + 2.f * t.m[i][j];
            g.v_t[i] += vp[i];
        }
    }
}

GS_DEV void grad_zero(ProjGrad &g) {
    g.v_px = g.v_py = g.v_pz = 0.f;
    g.v_S = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    g.v_W = mat3_zero();
    g.v_t[0] = g.v_t[1] = g.v_t[2] = 0.f;
}

} // namespace
