/*
 * gs_oracle.c -- CPU restatement of the reference's rasterize + quantize hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing in the product (gscodec_studio_amd/) may import,
 * link or call this file; it is used by tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py, always as the checker / baseline, never as the thing
 * shipped or measured as the product.
 *
 * Every function restates one reference kernel / torch function in plain fp32 C
 * (reference paths relative to the reference root):
 *   projection fwd/bwd : gsplat/cuda/csrc/fully_fused_projection_fwd.cu:22-196,
 *                        fully_fused_projection_bwd.cu:24-263,
 *                        include/{quat,quat_scale_to_covar_preci,transform,proj,utils}.cuh
 *                        (same maths as gsplat/cuda/_torch_impl.py:41-327)
 *   SH fwd/bwd         : gsplat/cuda/include/spherical_harmonics.cuh:13-362
 *                        (= gsplat/cuda/_torch_impl.py:620-714)
 *   isect count/emit   : gsplat/cuda/csrc/isect_tiles.cu:16-104 (= _torch_impl.py:331-399)
 *   stable sort        : semantics of cub::DeviceRadixSort::SortPairs (isect_tiles.cu:245-299)
 *   offset encode      : gsplat/cuda/csrc/isect_tiles.cu:308-354 (= _torch_impl.py:403-429)
 *   compositing fwd/bwd: gsplat/cuda/csrc/rasterize_to_pixels_fwd.cu:59-184,
 *                        rasterize_to_pixels_bwd.cu:105-275
 *   quantizers         : gsplat/compression_simulation/ops.py:39-75
 *
 * Pinning (see oracle/README.md, tests/golden/make_golden.py): projection, SH, isect,
 * offset-encode and the quantizers are checked against the reference's own Python
 * (torch) functions imported in the build container; compositing has NO runnable
 * reference there (it needs the CUDA extension + nerfacc), so for that stage parity is
 * pinned only by a dense autograd formulation and finite differences: "parity unpinned"
 * by executable reference code.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -fno-fast-math -shared -fPIC (oracle/Makefile).
 * Matrices are row-major m[r][c]; quaternions are (w,x,y,z).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

typedef struct { float m[3][3]; } M3;

static M3 m3_zero(void) { M3 r; memset(&r, 0, sizeof r); return r; }
static M3 m3_mul(M3 a, M3 b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
static M3 m3_T(M3 a) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
    return r;
}
static M3 m3_add(M3 a, M3 b) {
    M3 r;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][j] + b.m[i][j];
    return r;
}

/* quat.cuh:9-31 */
static M3 quat_to_rotmat(const float *q) {
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float inv = 1.0f / sqrtf(x * x + y * y + z * z + w * w);
    x *= inv; y *= inv; z *= inv; w *= inv;
    float x2 = x * x, y2 = y * y, z2 = z * z, xy = x * y, xz = x * z, yz = y * z, wx = w * x, wy = w * y, wz = w * z;
    M3 R;
    R.m[0][0] = 1.f - 2.f * (y2 + z2); R.m[0][1] = 2.f * (xy - wz); R.m[0][2] = 2.f * (xz + wy);
    R.m[1][0] = 2.f * (xy + wz); R.m[1][1] = 1.f - 2.f * (x2 + z2); R.m[1][2] = 2.f * (yz - wx);
    R.m[2][0] = 2.f * (xz - wy); R.m[2][1] = 2.f * (yz + wx); R.m[2][2] = 1.f - 2.f * (x2 + y2);
    return R;
}

/* quat.cuh:33-57 ; V[r][c] = d/dR[r][c] */
static void quat_to_rotmat_vjp(const float *q, M3 V, float *vq) {
    float w = q[0], x = q[1], y = q[2], z = q[3];
    float inv = 1.0f / sqrtf(x * x + y * y + z * z + w * w);
    x *= inv; y *= inv; z *= inv; w *= inv;
    float g[4];
    g[0] = 2.f * (x * (V.m[2][1] - V.m[1][2]) + y * (V.m[0][2] - V.m[2][0]) + z * (V.m[1][0] - V.m[0][1]));
    g[1] = 2.f * (-2.f * x * (V.m[1][1] + V.m[2][2]) + y * (V.m[1][0] + V.m[0][1]) + z * (V.m[2][0] + V.m[0][2]) + w * (V.m[2][1] - V.m[1][2]));
    g[2] = 2.f * (x * (V.m[1][0] + V.m[0][1]) - 2.f * y * (V.m[0][0] + V.m[2][2]) + z * (V.m[2][1] + V.m[1][2]) + w * (V.m[0][2] - V.m[2][0]));
    g[3] = 2.f * (x * (V.m[2][0] + V.m[0][2]) + y * (V.m[2][1] + V.m[1][2]) - 2.f * z * (V.m[0][0] + V.m[1][1]) + w * (V.m[1][0] - V.m[0][1]));
    float qn[4] = {w, x, y, z};
    float dot = g[0] * qn[0] + g[1] * qn[1] + g[2] * qn[2] + g[3] * qn[3];
    for (int i = 0; i < 4; ++i) vq[i] += (g[i] - dot * qn[i]) * inv;
}

/* quat_scale_to_covar_preci.cuh:10-41 */
static M3 covar_from_qs(const float *q, const float *s) {
    M3 R = quat_to_rotmat(q), M;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M.m[i][j] = R.m[i][j] * s[j];
    return m3_mul(M, m3_T(M));
}

/* quat_scale_to_covar_preci.cuh:43-81 */
static void covar_vjp_qs(const float *q, const float *s, M3 vC, float *vq, float *vs) {
    M3 R = quat_to_rotmat(q), M, vR;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M.m[i][j] = R.m[i][j] * s[j];
    M3 vM = m3_mul(m3_add(vC, m3_T(vC)), M);
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) vR.m[i][j] = vM.m[i][j] * s[j];
    quat_to_rotmat_vjp(q, vR, vq);
    for (int j = 0; j < 3; ++j) vs[j] += R.m[0][j] * vM.m[0][j] + R.m[1][j] * vM.m[1][j] + R.m[2][j] * vM.m[2][j];
}

typedef struct { float j[2][3]; } J23;

/* cov2d = J S J^T (2x2) */
static void proj_cov(J23 J, M3 S, float c[2][2]) {
    float JS[2][3];
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 3; ++k) JS[i][k] = J.j[i][0] * S.m[0][k] + J.j[i][1] * S.m[1][k] + J.j[i][2] * S.m[2][k];
    for (int i = 0; i < 2; ++i)
        for (int k = 0; k < 2; ++k) c[i][k] = JS[i][0] * J.j[k][0] + JS[i][1] * J.j[k][1] + JS[i][2] * J.j[k][2];
}

static void persp_limits(float fx, float fy, float cx, float cy, int W, int H, float *lxp, float *lxn, float *lyp, float *lyn) {
    float tan_fovx = 0.5f * W / fx, tan_fovy = 0.5f * H / fy;
    *lxp = (W - cx) / fx + 0.3f * tan_fovx;
    *lxn = cx / fx + 0.3f * tan_fovx;
    *lyp = (H - cy) / fy + 0.3f * tan_fovy;
    *lyn = cy / fy + 0.3f * tan_fovy;
}

/* proj.cuh:80-119 / 9-37 / 202-243 */
static J23 camera_jac(int model, const float *pc, float fx, float fy, float cx, float cy, int W, int H, float *m2, float *txy) {
    float x = pc[0], y = pc[1], z = pc[2];
    J23 J;
    memset(&J, 0, sizeof J);
    if (model == 0) {
        float lxp, lxn, lyp, lyn;
        persp_limits(fx, fy, cx, cy, W, H, &lxp, &lxn, &lyp, &lyn);
        float rz = 1.f / z, rz2 = rz * rz;
        float tx = z * fminf(lxp, fmaxf(-lxn, x * rz));
        float ty = z * fminf(lyp, fmaxf(-lyn, y * rz));
        J.j[0][0] = fx * rz; J.j[1][1] = fy * rz; J.j[0][2] = -fx * tx * rz2; J.j[1][2] = -fy * ty * rz2;
        m2[0] = fx * x * rz + cx; m2[1] = fy * y * rz + cy;
        if (txy) { txy[0] = tx; txy[1] = ty; }
    } else if (model == 1) {
        J.j[0][0] = fx; J.j[1][1] = fy;
        m2[0] = fx * x + cx; m2[1] = fy * y + cy;
    } else {
        float eps = 0.0000001f;
        float xy_len = sqrtf(x * x + y * y) + eps;
        float theta = atan2f(xy_len, z + eps);
        m2[0] = x * fx * theta / xy_len + cx;
        m2[1] = y * fy * theta / xy_len + cy;
        float x2 = x * x + eps, y2 = y * y, xy = x * y, x2y2 = x2 + y2;
        float inv = 1.f / (x2y2 + z * z);
        float b = atan2f(xy_len, z) / xy_len / x2y2;
        float a = z * inv / x2y2;
        J.j[0][0] = fx * (x2 * a + y2 * b); J.j[0][1] = fx * xy * (a - b); J.j[0][2] = -fx * x * inv;
        J.j[1][0] = fy * xy * (a - b); J.j[1][1] = fy * (y2 * a + x2 * b); J.j[1][2] = -fy * y * inv;
    }
    return J;
}

static void load_cam(const float *V, M3 *R, float *t) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) R->m[i][j] = V[4 * i + j];
        t[i] = V[4 * i + 3];
    }
}

static M3 load_covar(const float *covars, const float *quats, const float *scales, uint32_t n) {
    if (covars) {
        const float *c = covars + 6 * (size_t)n;
        M3 S = {{{c[0], c[1], c[2]}, {c[1], c[3], c[4]}, {c[2], c[4], c[5]}}};
        return S;
    }
    return covar_from_qs(quats + 4 * (size_t)n, scales + 3 * (size_t)n);
}

/* fully_fused_projection_fwd.cu:22-196; packed_formula selects packed_fwd.cu:183-186 */
void orc_projection_fwd(uint32_t C, uint32_t N, const float *means, const float *covars, const float *quats,
                        const float *scales, const float *viewmats, const float *Ks, int W, int H, float eps2d,
                        float near_plane, float far_plane, float radius_clip, int model, int packed_formula,
                        int32_t *radii, float *means2d, float *depths, float *conics, float *compensations) {
#pragma omp parallel for schedule(static)
    for (int64_t idx = 0; idx < (int64_t)C * N; ++idx) {
        uint32_t c = (uint32_t)(idx / N), n = (uint32_t)(idx % N);
        M3 R; float t[3];
        load_cam(viewmats + 16 * c, &R, t);
        const float *K = Ks + 9 * c;
        const float *p = means + 3 * (size_t)n;
        float pc[3];
        for (int i = 0; i < 3; ++i) pc[i] = R.m[i][0] * p[0] + R.m[i][1] * p[1] + R.m[i][2] * p[2] + t[i];
        radii[idx] = 0;
        if (pc[2] < near_plane || pc[2] > far_plane) continue;
        M3 S = load_covar(covars, quats, scales, n);
        M3 Sc = m3_mul(m3_mul(R, S), m3_T(R));
        float m2[2];
        J23 J = camera_jac(model, pc, K[0], K[4], K[2], K[5], W, H, m2, NULL);
        float c2[2][2];
        proj_cov(J, Sc, c2);
        /* utils.cuh:30-37 */
        float det_orig = c2[0][0] * c2[1][1] - c2[0][1] * c2[1][0];
        c2[0][0] += eps2d; c2[1][1] += eps2d;
        float det = c2[0][0] * c2[1][1] - c2[0][1] * c2[1][0];
        float comp = sqrtf(fmaxf(0.f, det_orig / det));
        if (det <= 0.f) continue;
        float inv_det = 1.f / det;
        float b = 0.5f * (c2[0][0] + c2[1][1]);
        float radius;
        if (packed_formula) {
            float v1 = b + sqrtf(fmaxf(0.1f, b * b - det)), v2 = b - sqrtf(fmaxf(0.1f, b * b - det));
            radius = ceilf(3.f * sqrtf(fmaxf(v1, v2)));
        } else {
            float v1 = b + sqrtf(fmaxf(0.01f, b * b - det));
            radius = ceilf(3.f * sqrtf(v1));
        }
        if (radius <= radius_clip) continue;
        if (m2[0] + radius <= 0 || m2[0] - radius >= W || m2[1] + radius <= 0 || m2[1] - radius >= H) continue;
        radii[idx] = (int32_t)radius;
        means2d[2 * idx] = m2[0]; means2d[2 * idx + 1] = m2[1];
        depths[idx] = pc[2];
        conics[3 * idx] = c2[1][1] * inv_det;
        conics[3 * idx + 1] = -c2[0][1] * inv_det;
        conics[3 * idx + 2] = c2[0][0] * inv_det;
        if (compensations) compensations[idx] = comp;
    }
}

/* fully_fused_projection_bwd.cu:24-263.  Outputs must be zero-initialised (the
 * reference accumulates with atomics); sums over cameras are done in double. */
void orc_projection_bwd(uint32_t C, uint32_t N, const float *means, const float *covars, const float *quats,
                        const float *scales, const float *viewmats, const float *Ks, int W, int H, float eps2d,
                        int model, const int32_t *radii, const float *conics, const float *compensations,
                        const float *v_means2d, const float *v_depths, const float *v_conics,
                        const float *v_compensations, float *v_means, float *v_covars, float *v_quats,
                        float *v_scales, float *v_viewmats) {
    double *vview = v_viewmats ? (double *)calloc((size_t)C * 16, sizeof(double)) : NULL;
    for (uint32_t n = 0; n < N; ++n) {
        double am[3] = {0, 0, 0}, ac[6] = {0, 0, 0, 0, 0, 0}, aq[4] = {0, 0, 0, 0}, as[3] = {0, 0, 0};
        for (uint32_t c = 0; c < C; ++c) {
            size_t idx = (size_t)c * N + n;
            if (radii[idx] <= 0) continue;
            const float *con = conics + 3 * idx, *vcn = v_conics + 3 * idx;
            /* inverse vjp: v_M = -P G P (utils.cuh:22-27) */
            float P[2][2] = {{con[0], con[1]}, {con[1], con[2]}};
            float G[2][2] = {{vcn[0], vcn[1] * .5f}, {vcn[1] * .5f, vcn[2]}};
            float PG[2][2], v2[2][2];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) PG[i][j] = P[i][0] * G[0][j] + P[i][1] * G[1][j];
            for (int i = 0; i < 2; ++i)
                for (int j = 0; j < 2; ++j) v2[i][j] = -(PG[i][0] * P[0][j] + PG[i][1] * P[1][j]);
            if (v_compensations) {
                /* utils.cuh:39-73 */
                float comp = compensations[idx], vcmp = v_compensations[idx];
                float det_conic = P[0][0] * P[1][1] - P[0][1] * P[1][0];
                float v_sqr = vcmp * 0.5f / (comp + 1e-6f);
                float om = 1.f - comp * comp;
                v2[0][0] += v_sqr * (om * P[0][0] - eps2d * det_conic);
                v2[0][1] += v_sqr * (om * P[0][1]);
                v2[1][0] += v_sqr * (om * P[1][0]);
                v2[1][1] += v_sqr * (om * P[1][1] - eps2d * det_conic);
            }
            M3 R; float t[3];
            load_cam(viewmats + 16 * c, &R, t);
            const float *K = Ks + 9 * c;
            float fx = K[0], cx = K[2], fy = K[4], cy = K[5];
            const float *p = means + 3 * (size_t)n;
            float pc[3];
            for (int i = 0; i < 3; ++i) pc[i] = R.m[i][0] * p[0] + R.m[i][1] * p[1] + R.m[i][2] * p[2] + t[i];
            M3 S = load_covar(covars, quats, scales, n);
            M3 Sc = m3_mul(m3_mul(R, S), m3_T(R));
            float m2[2], txy[2] = {0, 0};
            J23 J = camera_jac(model, pc, fx, fy, cx, cy, W, H, m2, txy);
            float x = pc[0], y = pc[1], z = pc[2];
            /* v_cov3d = J^T v2 J */
            M3 vSc = m3_zero();
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j)
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b) vSc.m[i][j] += J.j[a][i] * v2[a][b] * J.j[b][j];
            /* v_J = v2 J Sc^T + v2^T J Sc */
            float vJ[2][3];
            for (int a = 0; a < 2; ++a)
                for (int k = 0; k < 3; ++k) {
                    float s = 0.f;
                    for (int b = 0; b < 2; ++b)
                        for (int j = 0; j < 3; ++j) s += v2[a][b] * J.j[b][j] * Sc.m[k][j] + v2[b][a] * J.j[b][j] * Sc.m[j][k];
                    vJ[a][k] = s;
                }
            const float *vm = v_means2d + 2 * idx;
            float vpc[3] = {0, 0, 0};
            if (model == 0) {
                /* proj.cuh:122-199 */
                float lxp, lxn, lyp, lyn;
                persp_limits(fx, fy, cx, cy, W, H, &lxp, &lxn, &lyp, &lyn);
                float rz = 1.f / z, rz2 = rz * rz, rz3 = rz2 * rz, tx = txy[0], ty = txy[1];
                vpc[0] += fx * rz * vm[0];
                vpc[1] += fy * rz * vm[1];
                vpc[2] += -(fx * x * vm[0] + fy * y * vm[1]) * rz2;
                if (x * rz <= lxp && x * rz >= -lxn) vpc[0] += -fx * rz2 * vJ[0][2];
                else vpc[2] += -fx * rz3 * vJ[0][2] * tx;
                if (y * rz <= lyp && y * rz >= -lyn) vpc[1] += -fy * rz2 * vJ[1][2];
                else vpc[2] += -fy * rz3 * vJ[1][2] * ty;
                vpc[2] += -fx * rz2 * vJ[0][0] - fy * rz2 * vJ[1][1] + 2.f * fx * tx * rz3 * vJ[0][2] + 2.f * fy * ty * rz3 * vJ[1][2];
            } else if (model == 1) {
                /* proj.cuh:39-77 */
                vpc[0] += fx * vm[0];
                vpc[1] += fy * vm[1];
            } else {
                /* proj.cuh:245-343 */
                const float eps = 0.0000001f;
                float x2 = x * x + eps, y2 = y * y, xy = x * y, x2y2 = x2 + y2;
                float len_xy = sqrtf(x * x + y * y) + eps;
                float x2y2z2 = x2y2 + z * z, inv = 1.f / x2y2z2;
                float b = atan2f(len_xy, z) / len_xy / x2y2;
                float a = z * inv / x2y2;
                vpc[0] += fx * (x2 * a + y2 * b) * vm[0] + fy * xy * (a - b) * vm[1];
                vpc[1] += fx * xy * (a - b) * vm[0] + fy * (y2 * a + x2 * b) * vm[1];
                vpc[2] += -fx * x * inv * vm[0] - fy * y * inv * vm[1];
                float theta = atan2f(len_xy, z);
                float l4 = x2y2z2 * x2y2z2;
                float E = -l4 * x2y2 * theta + x2y2z2 * x2y2 * len_xy * z;
                float F = 3 * l4 * theta - 3 * x2y2z2 * len_xy * z - 2 * x2y2 * len_xy * z;
                float A = x * (3 * E + x2 * F), B = y * (E + x2 * F), Cc = x * (E + y2 * F), D = y * (3 * E + y2 * F);
                float S1 = x2 - y2 - z * z, S2 = y2 - x2 - z * z;
                float inv1 = inv * inv, inv2 = inv1 / (x2y2 * x2y2 * len_xy);
                float dx00 = fx * A * inv2, dx01 = fx * B * inv2, dx02 = fx * S1 * inv1;
                float dx10 = fy * B * inv2, dx11 = fy * Cc * inv2, dx12 = 2.f * fy * xy * inv1;
                float dy00 = dx01, dy01 = fx * Cc * inv2, dy02 = 2.f * fx * xy * inv1;
                float dy10 = dx11, dy11 = fy * D * inv2, dy12 = fy * S2 * inv1;
                float dz00 = dx02, dz01 = dy02, dz02 = 2.f * fx * x * z * inv1;
                float dz10 = dx12, dz11 = dy12, dz12 = 2.f * fy * y * z * inv1;
                vpc[0] += dx00 * vJ[0][0] + dx01 * vJ[0][1] + dx02 * vJ[0][2] + dx10 * vJ[1][0] + dx11 * vJ[1][1] + dx12 * vJ[1][2];
                vpc[1] += dy00 * vJ[0][0] + dy01 * vJ[0][1] + dy02 * vJ[0][2] + dy10 * vJ[1][0] + dy11 * vJ[1][1] + dy12 * vJ[1][2];
                vpc[2] += dz00 * vJ[0][0] + dz01 * vJ[0][1] + dz02 * vJ[0][2] + dz10 * vJ[1][0] + dz11 * vJ[1][1] + dz12 * vJ[1][2];
            }
            vpc[2] += v_depths[idx];
            /* transform.cuh:19-69 */
            for (int j = 0; j < 3; ++j) am[j] += R.m[0][j] * vpc[0] + R.m[1][j] * vpc[1] + R.m[2][j] * vpc[2];
            M3 vS = m3_mul(m3_mul(m3_T(R), vSc), R);
            if (vview) {
                M3 t1 = m3_mul(m3_mul(vSc, R), m3_T(S));
                M3 t2 = m3_mul(m3_mul(m3_T(vSc), R), S);
                for (int i = 0; i < 3; ++i) {
                    for (int j = 0; j < 3; ++j) vview[16 * c + 4 * i + j] += vpc[i] * p[j] + t1.m[i][j] + t2.m[i][j];
                    vview[16 * c + 4 * i + 3] += vpc[i];
                }
            }
            if (covars) {
                ac[0] += vS.m[0][0]; ac[1] += vS.m[0][1] + vS.m[1][0]; ac[2] += vS.m[0][2] + vS.m[2][0];
                ac[3] += vS.m[1][1]; ac[4] += vS.m[1][2] + vS.m[2][1]; ac[5] += vS.m[2][2];
            } else {
                float vq[4] = {0, 0, 0, 0}, vs[3] = {0, 0, 0};
                covar_vjp_qs(quats + 4 * (size_t)n, scales + 3 * (size_t)n, vS, vq, vs);
                for (int i = 0; i < 4; ++i) aq[i] += vq[i];
                for (int i = 0; i < 3; ++i) as[i] += vs[i];
            }
        }
        if (v_means) for (int i = 0; i < 3; ++i) v_means[3 * (size_t)n + i] = (float)am[i];
        if (v_covars) for (int i = 0; i < 6; ++i) v_covars[6 * (size_t)n + i] = (float)ac[i];
        if (v_quats) for (int i = 0; i < 4; ++i) v_quats[4 * (size_t)n + i] = (float)aq[i];
        if (v_scales) for (int i = 0; i < 3; ++i) v_scales[3 * (size_t)n + i] = (float)as[i];
    }
    if (vview) {
        for (size_t i = 0; i < (size_t)C * 16; ++i) v_viewmats[i] = (float)vview[i];
        free(vview);
    }
}

/* spherical_harmonics.cuh:13-101 (per element, all three channels) */
static void sh_bases(int degree, float x, float y, float z, float *Y) {
    Y[0] = 0.2820947917738781f;
    if (degree < 1) return;
    Y[1] = -0.48860251190292f * y; Y[2] = 0.48860251190292f * z; Y[3] = -0.48860251190292f * x;
    if (degree < 2) return;
    float z2 = z * z, fTmp0B = -1.092548430592079f * z, fC1 = x * x - y * y, fS1 = 2.f * x * y;
    Y[6] = 0.9461746957575601f * z2 - 0.3153915652525201f;
    Y[7] = fTmp0B * x; Y[5] = fTmp0B * y; Y[8] = 0.5462742152960395f * fC1; Y[4] = 0.5462742152960395f * fS1;
    if (degree < 3) return;
    float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f, fTmp1B = 1.445305721320277f * z;
    float fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
    Y[12] = z * (1.865881662950577f * z2 - 1.119528997770346f);
    Y[13] = fTmp0C * x; Y[11] = fTmp0C * y; Y[14] = fTmp1B * fC1; Y[10] = fTmp1B * fS1;
    Y[15] = -0.5900435899266435f * fC2; Y[9] = -0.5900435899266435f * fS2;
    if (degree < 4) return;
    float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f, fTmp2B = -1.770130769779931f * z;
    float fC3 = x * fC2 - y * fS2, fS3 = x * fS2 + y * fC2;
    Y[20] = 1.984313483298443f * z * Y[12] - 1.006230589874905f * Y[6];
    Y[21] = fTmp0D * x; Y[19] = fTmp0D * y; Y[22] = fTmp1C * fC1; Y[18] = fTmp1C * fS1;
    Y[23] = fTmp2B * fC2; Y[17] = fTmp2B * fS2; Y[24] = 0.6258357354491763f * fC3; Y[16] = 0.6258357354491763f * fS3;
}

void orc_sh_fwd(uint64_t n_elems, uint32_t K, uint32_t degree, const float *dirs, const float *coeffs,
                const uint8_t *masks, float *colors) {
    int nb = (degree + 1) * (degree + 1);
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < (int64_t)n_elems; ++e) {
        if (masks && !masks[e]) continue;
        float Y[25], x = 0, y = 0, z = 1;
        if (degree >= 1) {
            const float *d = dirs + 3 * e;
            float inorm = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            x = d[0] * inorm; y = d[1] * inorm; z = d[2] * inorm;
        }
        sh_bases(degree, x, y, z, Y);
        const float *cf = coeffs + (size_t)e * K * 3;
        for (int c = 0; c < 3; ++c) {
            float r = 0.f;
            for (int k = 0; k < nb; ++k) r += Y[k] * cf[3 * k + c];
            colors[3 * e + c] = r;
        }
    }
}

/* spherical_harmonics.cuh:104-362: v_coeffs (zero-initialised by the caller) and v_dirs.
 * The direction gradient is obtained from analytic partial derivatives of every basis
 * function, written out term by term as in the reference. */
static void sh_bases_grad(int degree, float x, float y, float z, float *Yx, float *Yy, float *Yz) {
    for (int k = 0; k < 25; ++k) Yx[k] = Yy[k] = Yz[k] = 0.f;
    if (degree < 1) return;
    Yy[1] = -0.48860251190292f; Yz[2] = 0.48860251190292f; Yx[3] = -0.48860251190292f;
    if (degree < 2) return;
    float z2 = z * z, fTmp0B = -1.092548430592079f * z, fC1 = x * x - y * y, fS1 = 2.f * x * y;
    float fTmp0B_z = -1.092548430592079f, fC1_x = 2.f * x, fC1_y = -2.f * y, fS1_x = 2.f * y, fS1_y = 2.f * x;
    Yz[6] = 2.f * 0.9461746957575601f * z;
    Yx[7] = fTmp0B; Yz[7] = fTmp0B_z * x; Yy[5] = fTmp0B; Yz[5] = fTmp0B_z * y;
    Yx[8] = 0.5462742152960395f * fC1_x; Yy[8] = 0.5462742152960395f * fC1_y;
    Yx[4] = 0.5462742152960395f * fS1_x; Yy[4] = 0.5462742152960395f * fS1_y;
    if (degree < 3) return;
    float fTmp0C = -2.285228997322329f * z2 + 0.4570457994644658f, fTmp1B = 1.445305721320277f * z;
    float fC2 = x * fC1 - y * fS1, fS2 = x * fS1 + y * fC1;
    float fTmp0C_z = -2.285228997322329f * 2.f * z, fTmp1B_z = 1.445305721320277f;
    float fC2_x = fC1 + x * fC1_x - y * fS1_x, fC2_y = x * fC1_y - fS1 - y * fS1_y;
    float fS2_x = fS1 + x * fS1_x + y * fC1_x, fS2_y = x * fS1_y + fC1 + y * fC1_y;
    float pSH12_z = 3.f * 1.865881662950577f * z2 - 1.119528997770346f;
    Yz[12] = pSH12_z;
    Yx[13] = fTmp0C; Yz[13] = fTmp0C_z * x; Yy[11] = fTmp0C; Yz[11] = fTmp0C_z * y;
    Yx[14] = fTmp1B * fC1_x; Yy[14] = fTmp1B * fC1_y; Yz[14] = fTmp1B_z * fC1;
    Yx[10] = fTmp1B * fS1_x; Yy[10] = fTmp1B * fS1_y; Yz[10] = fTmp1B_z * fS1;
    Yx[15] = -0.5900435899266435f * fC2_x; Yy[15] = -0.5900435899266435f * fC2_y;
    Yx[9] = -0.5900435899266435f * fS2_x; Yy[9] = -0.5900435899266435f * fS2_y;
    if (degree < 4) return;
    float fTmp0D = z * (-4.683325804901025f * z2 + 2.007139630671868f);
    float fTmp1C = 3.31161143515146f * z2 - 0.47308734787878f, fTmp2B = -1.770130769779931f * z;
    float pSH12 = z * (1.865881662950577f * z2 - 1.119528997770346f);
    float fTmp0D_z = 3.f * -4.683325804901025f * z2 + 2.007139630671868f;
    float fTmp1C_z = 2.f * 3.31161143515146f * z, fTmp2B_z = -1.770130769779931f;
    float fC3_x = fC2 + x * fC2_x - y * fS2_x, fC3_y = x * fC2_y - fS2 - y * fS2_y;
    float fS3_x = fS2 + y * fC2_x + x * fS2_x, fS3_y = x * fS2_y + fC2 + y * fC2_y;
    Yz[20] = 1.984313483298443f * (pSH12 + z * pSH12_z) + -1.006230589874905f * Yz[6];
    Yx[21] = fTmp0D; Yz[21] = fTmp0D_z * x; Yy[19] = fTmp0D; Yz[19] = fTmp0D_z * y;
    Yx[22] = fTmp1C * fC1_x; Yy[22] = fTmp1C * fC1_y; Yz[22] = fTmp1C_z * fC1;
    Yx[18] = fTmp1C * fS1_x; Yy[18] = fTmp1C * fS1_y; Yz[18] = fTmp1C_z * fS1;
    Yx[23] = fTmp2B * fC2_x; Yy[23] = fTmp2B * fC2_y; Yz[23] = fTmp2B_z * fC2;
    Yx[17] = fTmp2B * fS2_x; Yy[17] = fTmp2B * fS2_y; Yz[17] = fTmp2B_z * fS2;
    Yx[24] = 0.6258357354491763f * fC3_x; Yy[24] = 0.6258357354491763f * fC3_y;
    Yx[16] = 0.6258357354491763f * fS3_x; Yy[16] = 0.6258357354491763f * fS3_y;
}

void orc_sh_bwd(uint64_t n_elems, uint32_t K, uint32_t degree, const float *dirs, const float *coeffs,
                const uint8_t *masks, const float *v_colors, float *v_coeffs, float *v_dirs) {
    int nb = (degree + 1) * (degree + 1);
#pragma omp parallel for schedule(static)
    for (int64_t e = 0; e < (int64_t)n_elems; ++e) {
        if (masks && !masks[e]) continue;
        float Y[25], Yx[25], Yy[25], Yz[25], x = 0, y = 0, z = 1, inorm = 1;
        if (degree >= 1) {
            const float *d = dirs + 3 * e;
            inorm = 1.0f / sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            x = d[0] * inorm; y = d[1] * inorm; z = d[2] * inorm;
        }
        sh_bases(degree, x, y, z, Y);
        const float *cf = coeffs + (size_t)e * K * 3;
        float *vcf = v_coeffs + (size_t)e * K * 3;
        float vx = 0, vy = 0, vz = 0;
        if (v_dirs && degree >= 1) sh_bases_grad(degree, x, y, z, Yx, Yy, Yz);
        for (int c = 0; c < 3; ++c) {
            float vc = v_colors[3 * e + c];
            for (int k = 0; k < nb; ++k) vcf[3 * k + c] = Y[k] * vc;
            if (v_dirs && degree >= 1)
                for (int k = 1; k < nb; ++k) {
                    vx += vc * Yx[k] * cf[3 * k + c];
                    vy += vc * Yy[k] * cf[3 * k + c];
                    vz += vc * Yz[k] * cf[3 * k + c];
                }
        }
        if (v_dirs) {
            if (degree >= 1) {
                float dot = vx * x + vy * y + vz * z;
                v_dirs[3 * e] = (vx - dot * x) * inorm;
                v_dirs[3 * e + 1] = (vy - dot * y) * inorm;
                v_dirs[3 * e + 2] = (vz - dot * z) * inorm;
            } else {
                v_dirs[3 * e] = v_dirs[3 * e + 1] = v_dirs[3 * e + 2] = 0.f;
            }
        }
    }
}

/* isect_tiles.cu:56-75 */
static void tile_box(const float *m, int32_t radius, float ts, int tw, int th, int *x0, int *y0, int *x1, int *y1) {
    float tr = (float)radius / ts, tx = m[0] / ts, ty = m[1] / ts;
    float a;
    a = floorf(tx - tr); *x0 = a < 0 ? 0 : (a > tw ? tw : (int)a);
    a = floorf(ty - tr); *y0 = a < 0 ? 0 : (a > th ? th : (int)a);
    a = ceilf(tx + tr); *x1 = a < 0 ? 0 : (a > tw ? tw : (int)a);
    a = ceilf(ty + tr); *y1 = a < 0 ? 0 : (a > th ? th : (int)a);
}

/* pass 1: tiles_per_gauss; returns n_isects */
int64_t orc_isect_count(uint64_t n_elems, const float *means2d, const int32_t *radii, uint32_t tile_size,
                        uint32_t tw, uint32_t th, int32_t *tiles_per_gauss) {
    int64_t total = 0;
    for (uint64_t i = 0; i < n_elems; ++i) {
        int32_t cnt = 0;
        if (radii[i] > 0) {
            int x0, y0, x1, y1;
            tile_box(means2d + 2 * i, radii[i], (float)tile_size, (int)tw, (int)th, &x0, &y0, &x1, &y1);
            cnt = (y1 - y0) * (x1 - x0);
        }
        tiles_per_gauss[i] = cnt;
        total += cnt;
    }
    return total;
}

/* pass 2 (isect_tiles.cu:77-103); emission order = element order, row-major over the box */
void orc_isect_emit(uint64_t n_elems, uint32_t N, const int64_t *camera_ids, const float *means2d,
                    const int32_t *radii, const float *depths, uint32_t tile_size, uint32_t tw, uint32_t th,
                    uint32_t tile_n_bits, int64_t *isect_ids, int32_t *flatten_ids) {
    int64_t cur = 0;
    for (uint64_t i = 0; i < n_elems; ++i) {
        if (radii[i] <= 0) continue;
        int x0, y0, x1, y1;
        tile_box(means2d + 2 * i, radii[i], (float)tile_size, (int)tw, (int)th, &x0, &y0, &x1, &y1);
        int64_t cid = camera_ids ? camera_ids[i] : (int64_t)(i / N);
        int64_t cid_enc = cid << (32 + tile_n_bits);
        int32_t dbits;
        memcpy(&dbits, depths + i, 4);
        int64_t depth_enc = (int64_t)dbits;
        for (int y = y0; y < y1; ++y)
            for (int x = x0; x < x1; ++x) {
                int64_t tile_id = (int64_t)y * tw + x;
                isect_ids[cur] = cid_enc | (tile_id << 32) | depth_enc;
                flatten_ids[cur] = (int32_t)i;
                ++cur;
            }
    }
}

/* stable sort on key bits [0, end_bit): byte-wise LSD counting sort (obviously stable).
 * When end_bit == 64 the keys compare as SIGNED int64, as CUB does for int64_t keys. */
void orc_sort_pairs(uint64_t n, int64_t *keys, int32_t *vals, int end_bit) {
    if (n == 0) return;
    int64_t *k2 = (int64_t *)malloc(n * sizeof(int64_t));
    int32_t *v2 = (int32_t *)malloc(n * sizeof(int32_t));
    int64_t *src_k = keys, *dst_k = k2;
    int32_t *src_v = vals, *dst_v = v2;
    for (int shift = 0; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        uint32_t mask = (1u << bits) - 1u;
        uint32_t flip = (end_bit == 64 && shift + 8 >= 64) ? (1u << (bits - 1)) : 0u;
        uint64_t cnt[257];
        memset(cnt, 0, sizeof cnt);
        for (uint64_t i = 0; i < n; ++i) cnt[((((uint64_t)src_k[i] >> shift) & mask) ^ flip) + 1]++;
        for (int d = 0; d < 256; ++d) cnt[d + 1] += cnt[d];
        for (uint64_t i = 0; i < n; ++i) {
            uint32_t d = (uint32_t)((((uint64_t)src_k[i] >> shift) & mask) ^ flip);
            dst_k[cnt[d]] = src_k[i];
            dst_v[cnt[d]] = src_v[i];
            cnt[d]++;
        }
        int64_t *tk = src_k; src_k = dst_k; dst_k = tk;
        int32_t *tv = src_v; src_v = dst_v; dst_v = tv;
    }
    if (src_k != keys) {
        memcpy(keys, src_k, n * sizeof(int64_t));
        memcpy(vals, src_v, n * sizeof(int32_t));
    }
    free(k2);
    free(v2);
}

/* isect_tiles.cu:308-354 */
void orc_isect_offset_encode(uint64_t n_isects, const int64_t *isect_ids, uint32_t C, uint32_t n_tiles,
                             uint32_t tile_n_bits, int32_t *offsets) {
    uint64_t total = (uint64_t)C * n_tiles;
    if (n_isects == 0) { memset(offsets, 0, total * sizeof(int32_t)); return; }
    int64_t tmask = ((int64_t)1 << tile_n_bits) - 1;
    uint64_t next = 0; /* first tile slot not yet written */
    for (uint64_t idx = 0; idx < n_isects; ++idx) {
        int64_t cur = isect_ids[idx] >> 32;
        uint64_t id = (uint64_t)((cur >> tile_n_bits) * n_tiles + (cur & tmask));
        while (next <= id && next < total) offsets[next++] = (int32_t)idx;
    }
    while (next < total) offsets[next++] = (int32_t)n_isects;
}

/* rasterize_to_pixels_fwd.cu:59-184, per pixel (results do not depend on the batching).
 * borderline (optional, [C,H,W] uint8): set when a threshold decision of this pixel was
 * within a few ulp of flipping (alpha vs 1/255, next_T vs 1e-4, sigma vs 0) -- such pixels
 * legitimately differ between exp implementations and are excluded by the parity tests. */
void orc_rasterize_fwd(uint32_t C, uint32_t n_isects, uint32_t channels, const float *means2d,
                       const float *conics, const float *colors, const float *opacities,
                       const float *backgrounds, const uint8_t *masks, uint32_t W, uint32_t H,
                       uint32_t tile_size, uint32_t tw, uint32_t th, const int32_t *tile_offsets,
                       const int32_t *flatten_ids, float *render_colors, float *render_alphas,
                       int32_t *last_ids, uint8_t *borderline) {
    int64_t n_tiles_all = (int64_t)C * tw * th;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t tl = 0; tl < n_tiles_all; ++tl) {
        uint32_t cam = (uint32_t)(tl / (tw * th)), tile_id = (uint32_t)(tl % (tw * th));
        uint32_t ty = tile_id / tw, tx = tile_id % tw;
        int32_t rs = tile_offsets[tl];
        int32_t re = (tl == n_tiles_all - 1) ? (int32_t)n_isects : tile_offsets[tl + 1];
        const float *bg = backgrounds ? backgrounds + (size_t)cam * channels : NULL;
        float *pix_out = (float *)malloc(sizeof(float) * channels);
        for (uint32_t ly = 0; ly < tile_size; ++ly)
            for (uint32_t lx = 0; lx < tile_size; ++lx) {
                uint32_t i = ty * tile_size + ly, j = tx * tile_size + lx;
                if (i >= H || j >= W) continue;
                size_t pix = ((size_t)cam * H + i) * W + j;
                if (masks && !masks[tl]) {
                    for (uint32_t k = 0; k < channels; ++k) render_colors[pix * channels + k] = bg ? bg[k] : 0.f;
                    continue;
                }
                float px = (float)j + 0.5f, py = (float)i + 0.5f;
                float T = 1.f;
                uint32_t cur_idx = 0;
                uint8_t bl = 0;
                for (uint32_t k = 0; k < channels; ++k) pix_out[k] = 0.f;
                for (int32_t idx = rs; idx < re; ++idx) {
                    int32_t g = flatten_ids[idx];
                    const float *xy = means2d + 2 * (size_t)g, *con = conics + 3 * (size_t)g;
                    float opac = opacities[g];
                    float dx = xy[0] - px, dy = xy[1] - py;
                    float sigma = 0.5f * (con[0] * dx * dx + con[2] * dy * dy) + con[1] * dx * dy;
                    float alpha = fminf(0.999f, opac * expf(-sigma));
                    if (fabsf(alpha - 1.f / 255.f) < (1.f / 255.f) * 4e-6f * (1.f + fabsf(sigma))) bl = 1;
                    if (fabsf(sigma) < 1e-6f && alpha >= 1.f / 255.f) bl = 1;
                    if (sigma < 0.f || alpha < 1.f / 255.f) continue;
                    float next_T = T * (1.0f - alpha);
                    if (fabsf(next_T - 1e-4f) < 1e-4f * 3e-5f) bl = 1;
                    if (next_T <= 1e-4f) break;
                    float vis = alpha * T;
                    const float *c = colors + (size_t)g * channels;
                    for (uint32_t k = 0; k < channels; ++k) pix_out[k] += c[k] * vis;
                    cur_idx = (uint32_t)idx;
                    T = next_T;
                }
                render_alphas[pix] = 1.0f - T;
                for (uint32_t k = 0; k < channels; ++k) render_colors[pix * channels + k] = bg ? pix_out[k] + T * bg[k] : pix_out[k];
                last_ids[pix] = (int32_t)cur_idx;
                if (borderline) borderline[pix] = bl;
            }
        free(pix_out);
    }
}

/* Test helper (no reference counterpart): per pixel, the largest blending weight alpha_i * T_i that ANY entry of the
 * pixel's tile list could contribute, thresholds ignored (an entry just under alpha = 1/255 or with sigma just below 0
 * counts, and the walk continues past the stop).  When a threshold decision of a "borderline" pixel flips between two
 * exp implementations, ONE entry enters or leaves the sum: the colour moves by at most its own weight plus the rescaling
 * of everything behind it, i.e. by <= 2 * max_weight * max|colour|.  The parity tests bound the excluded pixels by it. */
void orc_rasterize_max_weight(uint32_t C, uint32_t n_isects, const float *means2d, const float *conics,
                              const float *opacities, uint32_t W, uint32_t H, uint32_t tile_size, uint32_t tw, uint32_t th,
                              const int32_t *tile_offsets, const int32_t *flatten_ids, float *max_weight) {
    int64_t n_tiles_all = (int64_t)C * tw * th;
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t tl = 0; tl < n_tiles_all; ++tl) {
        uint32_t cam = (uint32_t)(tl / (tw * th)), tile_id = (uint32_t)(tl % (tw * th));
        uint32_t ty = tile_id / tw, tx = tile_id % tw;
        int32_t rs = tile_offsets[tl];
        int32_t re = (tl == n_tiles_all - 1) ? (int32_t)n_isects : tile_offsets[tl + 1];
        for (uint32_t ly = 0; ly < tile_size; ++ly)
            for (uint32_t lx = 0; lx < tile_size; ++lx) {
                uint32_t i = ty * tile_size + ly, j = tx * tile_size + lx;
                if (i >= H || j >= W) continue;
                size_t pix = ((size_t)cam * H + i) * W + j;
                float px = (float)j + 0.5f, py = (float)i + 0.5f, T = 1.f, mw = 0.f;
                for (int32_t idx = rs; idx < re; ++idx) {
                    int32_t g = flatten_ids[idx];
                    const float *xy = means2d + 2 * (size_t)g, *con = conics + 3 * (size_t)g;
                    float dx = xy[0] - px, dy = xy[1] - py;
                    float sigma = 0.5f * (con[0] * dx * dx + con[2] * dy * dy) + con[1] * dx * dy;
                    float alpha = fminf(0.999f, opacities[g] * expf(-fmaxf(sigma, 0.f)));
                    mw = fmaxf(mw, alpha * T);
                    if (sigma < 0.f || alpha < 1.f / 255.f) continue;
                    T = fmaxf(T * (1.0f - alpha), 1e-4f); /* the stop may flip too: keep walking with the floor */
                }
                max_weight[pix] = mw;
            }
    }
}

/* rasterize_to_pixels_bwd.cu:105-275, per pixel back-to-front from last_ids.  The per-splat
 * sums (the reference's warp reductions + atomics) are accumulated in double and written
 * as fp32: gradient buffers are overwritten ([n_elems,*]). */
void orc_rasterize_bwd(uint32_t C, uint32_t n_elems, uint32_t n_isects, uint32_t channels, const float *means2d,
                       const float *conics, const float *colors, const float *opacities,
                       const float *backgrounds, const uint8_t *masks, uint32_t W, uint32_t H,
                       uint32_t tile_size, uint32_t tw, uint32_t th, const int32_t *tile_offsets,
                       const int32_t *flatten_ids, const float *render_alphas, const int32_t *last_ids,
                       const float *v_render_colors, const float *v_render_alphas, float *v_means2d_abs,
                       float *v_means2d, float *v_conics, float *v_colors, float *v_opacities) {
    size_t per = 2 + 3 + 1 + 2 + channels; /* xy, conic, opac, abs xy, colours */
    double *acc = (double *)calloc((size_t)n_elems * per, sizeof(double));
    int64_t n_tiles_all = (int64_t)C * tw * th;
    /* parallel over tiles; per-splat sums are double atomics (order-insensitive to ~1e-16) */
#pragma omp parallel for schedule(dynamic, 1)
    for (int64_t tl = 0; tl < n_tiles_all; ++tl) {
        if (masks && !masks[tl]) continue;
        uint32_t cam = (uint32_t)(tl / (tw * th)), tile_id = (uint32_t)(tl % (tw * th));
        uint32_t ty = tile_id / tw, tx = tile_id % tw;
        int32_t rs = tile_offsets[tl];
        int32_t re = (tl == n_tiles_all - 1) ? (int32_t)n_isects : tile_offsets[tl + 1];
        if (re <= rs) continue;
        const float *bg = backgrounds ? backgrounds + (size_t)cam * channels : NULL;
        float *buffer = (float *)malloc(sizeof(float) * channels);
        for (uint32_t ly = 0; ly < tile_size; ++ly)
            for (uint32_t lx = 0; lx < tile_size; ++lx) {
                uint32_t i = ty * tile_size + ly, j = tx * tile_size + lx;
                if (i >= H || j >= W) continue;
                size_t pix = ((size_t)cam * H + i) * W + j;
                float px = (float)j + 0.5f, py = (float)i + 0.5f;
                float T_final = 1.0f - render_alphas[pix];
                float T = T_final;
                int32_t bin_final = last_ids[pix];
                const float *vrc = v_render_colors + pix * channels;
                float vra = v_render_alphas ? v_render_alphas[pix] : 0.f;
                for (uint32_t k = 0; k < channels; ++k) buffer[k] = 0.f;
                for (int32_t idx = re - 1; idx >= rs; --idx) {
                    if (idx > bin_final) continue;
                    int32_t g = flatten_ids[idx];
                    const float *xy = means2d + 2 * (size_t)g, *con = conics + 3 * (size_t)g;
                    float opac = opacities[g];
                    float dx = xy[0] - px, dy = xy[1] - py;
                    float sigma = 0.5f * (con[0] * dx * dx + con[2] * dy * dy) + con[1] * dx * dy;
                    float vis = expf(-sigma);
                    float alpha = fminf(0.999f, opac * vis);
                    if (sigma < 0.f || alpha < 1.f / 255.f) continue;
                    float ra = 1.0f / (1.0f - alpha);
                    T *= ra;
                    float fac = alpha * T;
                    const float *c = colors + (size_t)g * channels;
                    double *a = acc + (size_t)g * per;
                    float v_alpha = 0.f;
                    for (uint32_t k = 0; k < channels; ++k) {
#pragma omp atomic
                        a[8 + k] += (double)(fac * vrc[k]);
                        v_alpha += (c[k] * T - buffer[k] * ra) * vrc[k];
                    }
                    v_alpha += T_final * ra * vra;
                    if (bg) {
                        float accum = 0.f;
                        for (uint32_t k = 0; k < channels; ++k) accum += bg[k] * vrc[k];
                        v_alpha += -T_final * ra * accum;
                    }
                    if (opac * vis <= 0.999f) {
                        float v_sigma = -opac * vis * v_alpha;
                        float vx = v_sigma * (con[0] * dx + con[1] * dy);
                        float vy = v_sigma * (con[1] * dx + con[2] * dy);
                        double add[8] = {vx, vy, 0.5f * v_sigma * dx * dx, v_sigma * dx * dy, 0.5f * v_sigma * dy * dy,
                                         vis * v_alpha, fabsf(vx), fabsf(vy)};
                        for (int q = 0; q < 8; ++q) {
#pragma omp atomic
                            a[q] += add[q];
                        }
                    }
                    for (uint32_t k = 0; k < channels; ++k) buffer[k] += c[k] * fac;
                }
            }
        free(buffer);
    }
    for (size_t g = 0; g < n_elems; ++g) {
        const double *a = acc + g * per;
        v_means2d[2 * g] = (float)a[0]; v_means2d[2 * g + 1] = (float)a[1];
        v_conics[3 * g] = (float)a[2]; v_conics[3 * g + 1] = (float)a[3]; v_conics[3 * g + 2] = (float)a[4];
        v_opacities[g] = (float)a[5];
        if (v_means2d_abs) { v_means2d_abs[2 * g] = (float)a[6]; v_means2d_abs[2 * g + 1] = (float)a[7]; }
        for (uint32_t k = 0; k < channels; ++k) v_colors[g * channels + k] = (float)a[8 + k];
    }
    free(acc);
}

/* ops.py:39-54 ("noise"): clamp, then + noise * q_step (two roundings, no fma) */
void orc_quant_noise_fwd(uint64_t n, const float *x, const float *noise, float lo, float hi, float q_step, float *out) {
    for (uint64_t i = 0; i < n; ++i) {
        float c = x[i] < lo ? lo : x[i];
        c = c > hi ? hi : c;
        float s = noise[i] * q_step;
        out[i] = c + s;
    }
}
void orc_quant_noise_bwd(uint64_t n, const float *x, const float *v_out, float lo, float hi, float *v_x) {
    for (uint64_t i = 0; i < n; ++i) v_x[i] = (x[i] >= lo && x[i] <= hi) ? v_out[i] : 0.f;
}
/* ops.py:57-75 (STE.forward): in-place clamp, normalise, round half to even, de-normalise */
void orc_quant_round_fwd(uint64_t n, float *x, float lo, float hi, float range, float qn, float *out) {
    for (uint64_t i = 0; i < n; ++i) {
        float c = x[i] < lo ? lo : x[i];
        c = c > hi ? hi : c;
        x[i] = c;
        float norm = (c - lo) / range;
        float lvl = rintf(norm / qn);
        float q = lvl * qn;
        float s = q * range;
        out[i] = s + lo;
    }
}
