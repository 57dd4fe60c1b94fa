// unfused.hip -- the reference's stand-alone public ops that the fused pipeline does not need but
// its API (and its tests/test_basic.py) exposes: world_to_cam, proj, rasterize_to_indices_in_range.
//
// Replaces gsplat/cuda/csrc/world_to_cam_{fwd,bwd}.cu, proj_{fwd,bwd}.cu and
// rasterize_to_indices_in_range.cu.  Unlike the fused projection these take GENERAL 3x3 matrices
// ([.., 3, 3] row-major, not assumed symmetric), exactly like the reference's glm code
// (include/transform.cuh:40-68, include/proj.cuh).  All three are streaming / helper kernels:
// one lane per element, no shared state; the backward of world_to_cam loops over cameras inside the
// lane (no atomics on the per-gaussian gradients) and reduces the per-camera pose gradient with DPP
// wave sums + one atomic group per wave.
#include "gs_common.h"
#include "proj_models.h"

namespace {

struct M3 {
    float m[3][3];
};

GS_DEV M3 load_m3(const float *p) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.m[i][j] = p[3 * i + j];
    return r;
}
GS_DEV void store_m3(float *p, const M3 &a) {
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) p[3 * i + j] = a.m[i][j];
}
GS_DEV M3 mul(const M3 &a, const M3 &b) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[i][0] * b.m[0][j] + a.m[i][1] * b.m[1][j] + a.m[i][2] * b.m[2][j];
    return r;
}
GS_DEV M3 transpose(const M3 &a) {
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.m[i][j] = a.m[j][i];
    return r;
}
GS_DEV M3 rot_of(const float *V) { // upper-left 3x3 of a row-major 4x4
    M3 r;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) r.m[i][j] = V[4 * i + j];
    return r;
}

// ---------------------------------------------------------------- world_to_cam
__global__ void __launch_bounds__(GS_BLOCK) world_to_cam_fwd_kernel(uint32_t C, uint32_t N, const float *__restrict__ means,
                                                                    const float *__restrict__ covars, const float *__restrict__ viewmats,
                                                                    float *__restrict__ means_c, float *__restrict__ covars_c) {
    const uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    const uint32_t c = blockIdx.y;
    if (n >= N) return;
    const float *V = viewmats + 16 * c; // wave-uniform
    const M3 R = rot_of(V);
    const size_t o = (size_t)c * N + n;
    if (means_c != nullptr) {
        const float *p = means + 3 * (size_t)n;
#pragma unroll
        for (int i = 0; i < 3; ++i) means_c[3 * o + i] = R.m[i][0] * p[0] + R.m[i][1] * p[1] + R.m[i][2] * p[2] + V[4 * i + 3];
    }
    if (covars_c != nullptr) {
        const M3 S = load_m3(covars + 9 * (size_t)n);
        store_m3(covars_c + 9 * o, mul(mul(R, S), transpose(R)));
    }
}

template <bool NEED_VIEW>
__global__ void __launch_bounds__(GS_BLOCK) world_to_cam_bwd_kernel(uint32_t C, uint32_t N, const float *__restrict__ means,
                                                                    const float *__restrict__ covars, const float *__restrict__ viewmats,
                                                                    const float *__restrict__ v_means_c, const float *__restrict__ v_covars_c,
                                                                    float *__restrict__ v_means, float *__restrict__ v_covars,
                                                                    float *__restrict__ v_viewmats) {
    const uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    const bool live = n < N;
    float p[3] = {0.f, 0.f, 0.f};
    M3 S = {};
    if (live) {
#pragma unroll
        for (int i = 0; i < 3; ++i) p[i] = means[3 * (size_t)n + i];
        if (covars != nullptr) S = load_m3(covars + 9 * (size_t)n);
    }
    float vp[3] = {0.f, 0.f, 0.f};
    M3 vS = {};
    for (uint32_t c = 0; c < C; ++c) {
        const float *V = viewmats + 16 * c;
        const M3 R = rot_of(V);
        const size_t o = (size_t)c * N + n;
        float g[3] = {0.f, 0.f, 0.f};
        M3 G = {};
        if (live && v_means_c != nullptr) {
#pragma unroll
            for (int i = 0; i < 3; ++i) g[i] = v_means_c[3 * o + i];
        }
        if (live && v_covars_c != nullptr) G = load_m3(v_covars_c + 9 * o);
        // v_p += R^T g ; v_S += R^T G R   (transform.cuh:19-37, 49-68)
#pragma unroll
        for (int j = 0; j < 3; ++j) vp[j] += R.m[0][j] * g[0] + R.m[1][j] * g[1] + R.m[2][j] * g[2];
        const M3 t = mul(mul(transpose(R), G), R);
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) vS.m[i][j] += t.m[i][j];
        if (NEED_VIEW) {
            // v_R = g p^T + G R S^T + G^T R S ; v_t = g   -- summed over the gaussians of this wave
            const M3 a = mul(mul(G, R), transpose(S));
            const M3 b = mul(mul(transpose(G), R), S);
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const float tot = wave_sum(g[i] * p[j] + a.m[i][j] + b.m[i][j]);
                    if (lane_id() == 0 && tot != 0.f) unsafeAtomicAdd(v_viewmats + 16 * c + 4 * i + j, tot);
                }
                const float tt = wave_sum(g[i]);
                if (lane_id() == 0 && tt != 0.f) unsafeAtomicAdd(v_viewmats + 16 * c + 4 * i + 3, tt);
            }
        }
    }
    if (!live) return;
    if (v_means != nullptr) {
#pragma unroll
        for (int i = 0; i < 3; ++i) v_means[3 * (size_t)n + i] = vp[i];
    }
    if (v_covars != nullptr) store_m3(v_covars + 9 * (size_t)n, vS);
}

// ---------------------------------------------------------------- proj
GS_DEV Camera intrinsics_only(const float *__restrict__ Ks, uint32_t c) {
    const float *K = Ks + 9 * c;
    Camera cam = {};
    cam.fx = K[0]; cam.cx = K[2]; cam.fy = K[4]; cam.cy = K[5];
    return cam;
}

GS_DEV void model_jac(const Camera &cam, int camera_model, float x, float y, float z, int W, int H, Jac &J, float &mx, float &my) {
    float txc, tyc;
    if (camera_model == GS_CAMERA_PINHOLE) pinhole_jac(cam, x, y, z, W, H, J, mx, my, txc, tyc);
    else if (camera_model == GS_CAMERA_ORTHO) ortho_jac(cam, x, y, J, mx, my);
    else fisheye_jac(cam, x, y, z, J, mx, my);
}

__global__ void __launch_bounds__(GS_BLOCK) proj_fwd_kernel(uint32_t C, uint32_t N, const float *__restrict__ means,
                                                            const float *__restrict__ covars, const float *__restrict__ Ks, int W, int H,
                                                            int camera_model, float *__restrict__ means2d, float *__restrict__ covars2d) {
    const uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    const uint32_t c = blockIdx.y;
    if (n >= N) return;
    const Camera cam = intrinsics_only(Ks, c);
    const size_t o = (size_t)c * N + n;
    const float *p = means + 3 * o;
    const M3 S = load_m3(covars + 9 * o);
    Jac J;
    float mx, my;
    model_jac(cam, camera_model, p[0], p[1], p[2], W, H, J, mx, my);
    const float Jm[2][3] = {{J.j00, J.j01, J.j02}, {J.j10, J.j11, J.j12}};
    float JS[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int l = 0; l < 3; ++l) JS[i][l] = Jm[i][0] * S.m[0][l] + Jm[i][1] * S.m[1][l] + Jm[i][2] * S.m[2][l];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) covars2d[4 * o + 2 * i + j] = JS[i][0] * Jm[j][0] + JS[i][1] * Jm[j][1] + JS[i][2] * Jm[j][2];
    means2d[2 * o] = mx;
    means2d[2 * o + 1] = my;
}

__global__ void __launch_bounds__(GS_BLOCK) proj_bwd_kernel(uint32_t C, uint32_t N, const float *__restrict__ means,
                                                            const float *__restrict__ covars, const float *__restrict__ Ks, int W, int H,
                                                            int camera_model, const float *__restrict__ v_means2d,
                                                            const float *__restrict__ v_covars2d, float *__restrict__ v_means,
                                                            float *__restrict__ v_covars) {
    const uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    const uint32_t c = blockIdx.y;
    if (n >= N) return;
    const Camera cam = intrinsics_only(Ks, c);
    const size_t o = (size_t)c * N + n;
    const float *p = means + 3 * o;
    const M3 S = load_m3(covars + 9 * o);
    Jac J;
    float mx, my;
    model_jac(cam, camera_model, p[0], p[1], p[2], W, H, J, mx, my);
    const float Jm[2][3] = {{J.j00, J.j01, J.j02}, {J.j10, J.j11, J.j12}};
    const float G[2][2] = {{v_covars2d[4 * o], v_covars2d[4 * o + 1]}, {v_covars2d[4 * o + 2], v_covars2d[4 * o + 3]}};
    // v_S = J^T G J
    float GJ[2][3], GtJ[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int l = 0; l < 3; ++l) {
            GJ[i][l] = G[i][0] * Jm[0][l] + G[i][1] * Jm[1][l];
            GtJ[i][l] = G[0][i] * Jm[0][l] + G[1][i] * Jm[1][l];
        }
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
        for (int l = 0; l < 3; ++l) v_covars[9 * o + 3 * k + l] = Jm[0][k] * GJ[0][l] + Jm[1][k] * GJ[1][l];
    // v_J = G J S^T + G^T J S   (proj.cuh:160-165)
    float vJ[2][3];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 3; ++k)
            vJ[i][k] = GJ[i][0] * S.m[k][0] + GJ[i][1] * S.m[k][1] + GJ[i][2] * S.m[k][2] +
                       GtJ[i][0] * S.m[0][k] + GtJ[i][1] * S.m[1][k] + GtJ[i][2] * S.m[2][k];
    float vx, vy, vz;
    proj_mean_vjp(cam, camera_model, p[0], p[1], p[2], W, H, v_means2d[2 * o], v_means2d[2 * o + 1], vJ[0][0], vJ[0][1], vJ[0][2],
                  vJ[1][0], vJ[1][1], vJ[1][2], vx, vy, vz);
    v_means[3 * o] = vx;
    v_means[3 * o + 1] = vy;
    v_means[3 * o + 2] = vz;
}

// ---------------------------------------------------------------- rasterize_to_indices_in_range
// One thread per pixel, one workgroup per tile, entries staged through LDS in batches of
// tile_size^2 (the batch size is part of the op's contract: range_start/range_end count batches).
// FILL = false: count the contributing splats per pixel; FILL = true: write (gaussian, pixel) pairs.
struct IdxRec {
    float x, y, opac, ca, cb, cc;
    int32_t g;
};

template <bool FILL>
__global__ void indices_in_range_kernel(uint32_t range_start, uint32_t range_end, uint32_t C, uint32_t N, uint32_t n_isects,
                                        const float *__restrict__ means2d, const float *__restrict__ conics,
                                        const float *__restrict__ opacities, uint32_t W, uint32_t H, uint32_t tile_size,
                                        uint32_t tile_width, uint32_t tile_height, const int32_t *__restrict__ tile_offsets,
                                        const int32_t *__restrict__ flatten_ids, const float *__restrict__ transmittances,
                                        const int32_t *__restrict__ chunk_starts, int32_t *__restrict__ chunk_cnts,
                                        int64_t *__restrict__ gaussian_ids, int64_t *__restrict__ pixel_ids) {
    extern __shared__ IdxRec s_batch[];
    const uint32_t cam = blockIdx.x;
    const uint32_t tile_id = blockIdx.y * tile_width + blockIdx.z;
    const uint32_t i = blockIdx.y * tile_size + threadIdx.y, jx = blockIdx.z * tile_size + threadIdx.x;
    const uint32_t block_size = blockDim.x * blockDim.y;
    const uint32_t tr = threadIdx.y * blockDim.x + threadIdx.x;
    const uint32_t lin = cam * tile_width * tile_height + tile_id;
    const float px = (float)jx + 0.5f, py = (float)i + 0.5f;
    const bool inside = i < H && jx < W;
    const size_t pix = (size_t)cam * H * W + (size_t)i * W + jx;
    bool done = !inside;
    const int32_t rs = tile_offsets[lin];
    const int32_t re = (lin + 1 == C * tile_width * tile_height) ? (int32_t)n_isects : tile_offsets[lin + 1];
    const uint32_t num_batches = ((uint32_t)(re - rs) + block_size - 1) / block_size;
    if (range_start >= num_batches) return; // this tile was finished by earlier calls (count stays at its zero fill)
    float trans = inside ? transmittances[pix] : 0.f;
    int32_t base = (FILL && inside) ? chunk_starts[pix] : 0;
    int32_t cnt = 0;
    const uint32_t b_end = range_end < num_batches ? range_end : num_batches;
    for (uint32_t b = range_start; b < b_end; ++b) {
        if (__syncthreads_count(done) >= (int)block_size) break;
        const uint32_t batch_start = (uint32_t)rs + block_size * b;
        const uint32_t idx = batch_start + tr;
        if (idx < (uint32_t)re) {
            const int32_t g = flatten_ids[idx];
            IdxRec r;
            r.g = g;
            r.x = means2d[2 * (size_t)g];
            r.y = means2d[2 * (size_t)g + 1];
            r.opac = opacities[g];
            r.ca = conics[3 * (size_t)g];
            r.cb = conics[3 * (size_t)g + 1];
            r.cc = conics[3 * (size_t)g + 2];
            s_batch[tr] = r;
        }
        __syncthreads();
        const uint32_t batch_size = min(block_size, (uint32_t)re - batch_start);
        for (uint32_t t = 0; t < batch_size && !done; ++t) {
            const IdxRec r = s_batch[t];
            const float dx = r.x - px, dy = r.y - py;
            const float sigma = 0.5f * (r.ca * dx * dx + r.cc * dy * dy) + r.cb * dx * dy;
            const float alpha = fminf(0.999f, r.opac * __expf(-sigma));
            if (sigma < 0.f || alpha < 1.f / 255.f) continue;
            const float next_trans = trans * (1.f - alpha);
            if (next_trans <= 1e-4f) { // exclusive stop
                done = true;
                break;
            }
            if (FILL) {
                gaussian_ids[base + cnt] = (int64_t)(r.g % (int32_t)N);
                pixel_ids[base + cnt] = (int64_t)pix;
            }
            cnt += 1;
            trans = next_trans;
        }
    }
    if (!FILL && inside) chunk_cnts[pix] = cnt;
}

} // namespace

// ---------------------------------------------------------------------------------------------------------------------
// Accumulated depth -> expected depth: the image-level tail of rasterization()'s "ED" / "RGB+ED" modes (reference
// gsplat/rendering.py:471-477):  out = cat(render[..., :-1], render[..., -1:] / alphas.clamp(min=1e-10)).  One lane per pixel; the
// backward is torch's: d/d render_last = v / a_c, d/d alpha = -v ((render_last / a_c) / a_c) where alpha >= 1e-10 (the clamp passes the
// gradient there), 0 below.
namespace {

constexpr float ED_ALPHA_MIN = 1e-10f;

__global__ void __launch_bounds__(GS_BLOCK) expected_depth_fwd_kernel(uint64_t n_pix, uint32_t ch, const float *__restrict__ renders,
                                                                      const float *__restrict__ alphas, float *__restrict__ out) {
    GS_FP_STRICT;
    const uint64_t p = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    if (p >= n_pix) return;
    const float a = fmaxf(alphas[p], ED_ALPHA_MIN);
    const float *r = renders + p * ch;
    float *o = out + p * ch;
    for (uint32_t k = 0; k + 1u < ch; ++k) o[k] = r[k];
    o[ch - 1u] = (r[ch - 1u] / a);
}

__global__ void __launch_bounds__(GS_BLOCK) expected_depth_bwd_kernel(uint64_t n_pix, uint32_t ch, const float *__restrict__ renders,
                                                                      const float *__restrict__ alphas, const float *__restrict__ v_out,
                                                                      float *__restrict__ v_renders, float *__restrict__ v_alphas) {
    GS_FP_STRICT;
    const uint64_t p = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    if (p >= n_pix) return;
    const float a0 = alphas[p], a = fmaxf(a0, ED_ALPHA_MIN);
    const float *v = v_out + p * ch;
    const float vd = v[ch - 1u];
    if (v_renders != nullptr) {
        float *o = v_renders + p * ch;
        for (uint32_t k = 0; k + 1u < ch; ++k) o[k] = v[k];
        o[ch - 1u] = (vd / a);
    }
    if (v_alphas != nullptr) v_alphas[p] = a0 >= ED_ALPHA_MIN ? (-vd * ((renders[p * ch + ch - 1u] / a) / a)) : 0.f;
}

} // namespace

extern "C" int32_t gs_expected_depth_fwd(uint64_t n_pix, uint32_t channels, const float *renders, const float *alphas, float *out,
                                         gs_stream_t stream) {
    if (n_pix == 0) return 0;
    GS_CHECK_ARG(renders && alphas && out && channels >= 1, "null pointer / no channels");
    GS_CHECK_ARG(n_pix < ((uint64_t)1 << 32) * GS_BLOCK, "too many pixels");
    hipLaunchKernelGGL(expected_depth_fwd_kernel, dim3((uint32_t)((n_pix + GS_BLOCK - 1) / GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream,
                       n_pix, channels, renders, alphas, out);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_expected_depth_bwd(uint64_t n_pix, uint32_t channels, const float *renders, const float *alphas, const float *v_out,
                                         float *v_renders, float *v_alphas, gs_stream_t stream) {
    if (n_pix == 0) return 0;
    GS_CHECK_ARG(renders && alphas && v_out && channels >= 1, "null pointer / no channels");
    GS_CHECK_ARG(n_pix < ((uint64_t)1 << 32) * GS_BLOCK, "too many pixels");
    hipLaunchKernelGGL(expected_depth_bwd_kernel, dim3((uint32_t)((n_pix + GS_BLOCK - 1) / GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream,
                       n_pix, channels, renders, alphas, v_out, v_renders, v_alphas);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_world_to_cam_fwd(uint32_t C, uint32_t N, const float *means, const float *covars, const float *viewmats,
                                       float *means_c, float *covars_c, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(viewmats != nullptr, "null viewmats");
    GS_CHECK_ARG((means_c == nullptr || means != nullptr) && (covars_c == nullptr || covars != nullptr), "output without its input");
    hipLaunchKernelGGL(world_to_cam_fwd_kernel, dim3(gs_div_up(N, GS_BLOCK), C), dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N, means,
                       covars, viewmats, means_c, covars_c);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_world_to_cam_bwd(uint32_t C, uint32_t N, const float *means, const float *covars, const float *viewmats,
                                       const float *v_means_c, const float *v_covars_c, float *v_means, float *v_covars,
                                       float *v_viewmats, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(means != nullptr && viewmats != nullptr, "null pointer");
    GS_CHECK_ARG(v_covars_c == nullptr || covars != nullptr, "v_covars_c without covars");
    dim3 grid(gs_div_up(N, GS_BLOCK));
    if (v_viewmats != nullptr)
        hipLaunchKernelGGL((world_to_cam_bwd_kernel<true>), grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N, means, covars, viewmats,
                           v_means_c, v_covars_c, v_means, v_covars, v_viewmats);
    else
        hipLaunchKernelGGL((world_to_cam_bwd_kernel<false>), grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N, means, covars, viewmats,
                           v_means_c, v_covars_c, v_means, v_covars, v_viewmats);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_proj_fwd(uint32_t C, uint32_t N, const float *means, const float *covars, const float *Ks, int32_t width,
                               int32_t height, int32_t camera_model, float *means2d, float *covars2d, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(means && covars && Ks && means2d && covars2d, "null pointer");
    GS_CHECK_ARG(camera_model >= 0 && camera_model <= 2, "unknown camera model");
    hipLaunchKernelGGL(proj_fwd_kernel, dim3(gs_div_up(N, GS_BLOCK), C), dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N, means, covars, Ks,
                       width, height, camera_model, means2d, covars2d);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_proj_bwd(uint32_t C, uint32_t N, const float *means, const float *covars, const float *Ks, int32_t width,
                               int32_t height, int32_t camera_model, const float *v_means2d, const float *v_covars2d, float *v_means,
                               float *v_covars, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(means && covars && Ks && v_means2d && v_covars2d && v_means && v_covars, "null pointer");
    GS_CHECK_ARG(camera_model >= 0 && camera_model <= 2, "unknown camera model");
    hipLaunchKernelGGL(proj_bwd_kernel, dim3(gs_div_up(N, GS_BLOCK), C), dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N, means, covars, Ks,
                       width, height, camera_model, v_means2d, v_covars2d, v_means, v_covars);
    GS_CHECK_LAUNCH();
    return 0;
}

static int32_t indices_launch(bool fill, uint32_t range_start, uint32_t range_end, uint32_t C, uint32_t N, uint32_t n_isects,
                              const float *means2d, const float *conics, const float *opacities, uint32_t W, uint32_t H,
                              uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, const int32_t *tile_offsets,
                              const int32_t *flatten_ids, const float *transmittances, const int32_t *chunk_starts, int32_t *chunk_cnts,
                              int64_t *gaussian_ids, int64_t *pixel_ids, hipStream_t st) {
    dim3 threads(tile_size, tile_size, 1), blocks(C, tile_height, tile_width);
    const size_t shmem = (size_t)tile_size * tile_size * sizeof(IdxRec);
    if (fill)
        hipLaunchKernelGGL((indices_in_range_kernel<true>), blocks, threads, shmem, st, range_start, range_end, C, N, n_isects, means2d,
                           conics, opacities, W, H, tile_size, tile_width, tile_height, tile_offsets, flatten_ids, transmittances,
                           chunk_starts, chunk_cnts, gaussian_ids, pixel_ids);
    else
        hipLaunchKernelGGL((indices_in_range_kernel<false>), blocks, threads, shmem, st, range_start, range_end, C, N, n_isects, means2d,
                           conics, opacities, W, H, tile_size, tile_width, tile_height, tile_offsets, flatten_ids, transmittances,
                           chunk_starts, chunk_cnts, gaussian_ids, pixel_ids);
    return 0;
}

extern "C" int32_t gs_rasterize_indices_count(uint32_t range_start, uint32_t range_end, uint32_t C, uint32_t N, uint32_t n_isects,
                                              const float *means2d, const float *conics, const float *opacities, uint32_t image_width,
                                              uint32_t image_height, uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
                                              const int32_t *tile_offsets, const int32_t *flatten_ids, const float *transmittances,
                                              int32_t *chunk_cnts, gs_stream_t stream) {
    if (C == 0 || n_isects == 0) return 0;
    GS_CHECK_ARG(means2d && conics && opacities && tile_offsets && flatten_ids && transmittances && chunk_cnts, "null pointer");
    GS_CHECK_ARG(tile_size >= 1 && tile_size <= 32, "tile_size must be in 1..32");
    indices_launch(false, range_start, range_end, C, N, n_isects, means2d, conics, opacities, image_width, image_height, tile_size,
                   tile_width, tile_height, tile_offsets, flatten_ids, transmittances, nullptr, chunk_cnts, nullptr, nullptr,
                   (hipStream_t)stream);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_rasterize_indices_fill(uint32_t range_start, uint32_t range_end, uint32_t C, uint32_t N, uint32_t n_isects,
                                             const float *means2d, const float *conics, const float *opacities, uint32_t image_width,
                                             uint32_t image_height, uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
                                             const int32_t *tile_offsets, const int32_t *flatten_ids, const float *transmittances,
                                             const int32_t *chunk_starts, int64_t *gaussian_ids, int64_t *pixel_ids, gs_stream_t stream) {
    if (C == 0 || n_isects == 0) return 0;
    GS_CHECK_ARG(means2d && conics && opacities && tile_offsets && flatten_ids && transmittances && chunk_starts && gaussian_ids &&
                     pixel_ids, "null pointer");
    GS_CHECK_ARG(tile_size >= 1 && tile_size <= 32, "tile_size must be in 1..32");
    indices_launch(true, range_start, range_end, C, N, n_isects, means2d, conics, opacities, image_width, image_height, tile_size,
                   tile_width, tile_height, tile_offsets, flatten_ids, transmittances, chunk_starts, nullptr, gaussian_ids, pixel_ids,
                   (hipStream_t)stream);
    GS_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------- accumulate
// The reference's `accumulate` (gsplat/cuda/_torch_impl.py:432-519): alpha compositing over an explicit list of M (gaussian, pixel,
// camera) intersections -- what rasterize_to_indices_in_range emits: grouped by ray (camera, pixel), front to back inside a ray.  The
// reference evaluates it with torch ops + nerfacc (absent from the tree and unpinned: examples/requirements.txt:9-10 names the
// git head; its two functions are restated from their published definitions):
//     alpha_m   = min(opacity exp(-sigma_m), 0.999)                                   (_torch_impl.py:490-501)
//     weight_m  = alpha_m prod_{k < m, same ray} (1 - alpha_k)                          nerfacc.render_weight_from_alpha
//     renders[ray] = sum_m weight_m colors_m ;  alphas[ray] = sum_m weight_m           nerfacc.accumulate_along_rays (index_add)
// A ray = a maximal run of consecutive entries with the same (camera, pixel) (nerfacc's packed_info over sorted ray_indices).
// One lane per intersection; the lane that starts a run walks it (runs are a few hundred entries at most; this is the reference's
// "playground" op, not the hot path).  Forward: grid.y = 1 + ceil(channels / 4): slice 0 writes weights and the alpha image, slice
// 1 + k four colour channels.  Backward: the run heads walk back to front (T_m = T_{m+1} / (1 - alpha_m)), leaving d L / d alpha_m;
// a second launch, one lane per (intersection, channel slice), scatters the parameter gradients with float atomics.
namespace {

struct AccArgs {
    uint64_t M;
    uint32_t C, N, channels;
    int32_t W, H;
    const float *means2d, *conics, *opacities, *colors; // [C,N,2] [C,N,3] [C,N] [C,N,channels]
    const int64_t *gids, *pids, *cids;                  // [M]
};

GS_DEV int64_t acc_ray(const AccArgs &a, uint64_t m) { return a.cids[m] * (int64_t)a.H * a.W + a.pids[m]; }

// alpha of entry m (and the pieces its gradient needs)
struct AccAlpha {
    float alpha, raw, dx, dy, e; // clamped | opacity exp(-sigma) | pixel - mean | exp(-sigma)
    size_t elem;                 // camera * N + gaussian
};
GS_DEV AccAlpha acc_alpha(const AccArgs &a, uint64_t m) {
    AccAlpha r;
    r.elem = (size_t)a.cids[m] * a.N + (size_t)a.gids[m];
    const int64_t p = a.pids[m];
    const float px = (float)(p % a.W) + 0.5f, py = (float)(p / a.W) + 0.5f;
    r.dx = px - a.means2d[2 * r.elem];
    r.dy = py - a.means2d[2 * r.elem + 1];
    const float ca = a.conics[3 * r.elem], cb = a.conics[3 * r.elem + 1], cc = a.conics[3 * r.elem + 2];
    const float sigma = 0.5f * (ca * r.dx * r.dx + cc * r.dy * r.dy) + cb * r.dx * r.dy;
    r.e = expf(-sigma);
    r.raw = a.opacities[r.elem] * r.e;
    r.alpha = fminf(r.raw, 0.999f);
    return r;
}

__global__ void __launch_bounds__(GS_BLOCK) accumulate_fwd_kernel(AccArgs a, float *__restrict__ alpha_buf, float *__restrict__ weights,
                                                                  float *__restrict__ renders, float *__restrict__ alphas) {
    const uint64_t m = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    if (m >= a.M) return;
    const int64_t ray = acc_ray(a, m);
    const uint32_t slice = blockIdx.y;
    if (slice == 0) alpha_buf[m] = acc_alpha(a, m).alpha; // (every lane: the backward and the other slices read it back)
    if (m > 0 && acc_ray(a, m - 1) == ray) return;          // not the head of its run
    float T = 1.f;
    if (slice == 0) {
        float acc = 0.f;
        for (uint64_t j = m; j < a.M && acc_ray(a, j) == ray; ++j) {
            const float al = acc_alpha(a, j).alpha;
            const float w = al * T;
            weights[j] = w;
            acc += w;
            T *= 1.f - al;
        }
        atomicAdd(alphas + ray, acc); // (index_add semantics: a ray split into several runs -- unsorted input -- sums its runs)
        return;
    }
    const uint32_t c0 = (slice - 1u) * 4u;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    for (uint64_t j = m; j < a.M && acc_ray(a, j) == ray; ++j) {
        const AccAlpha r = acc_alpha(a, j);
        const float w = r.alpha * T;
        const float *col = a.colors + r.elem * a.channels;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            if (c0 + k < a.channels) s[k] += w * col[c0 + k];
        T *= 1.f - r.alpha;
    }
#pragma unroll
    for (uint32_t k = 0; k < 4u; ++k)
        if (c0 + k < a.channels) atomicAdd(renders + ray * a.channels + c0 + k, s[k]);
}

// d L / d alpha_m for every entry (run heads walk their run back to front)
__global__ void __launch_bounds__(GS_BLOCK) accumulate_bwd_alpha_kernel(AccArgs a, const float *__restrict__ alpha_buf,
                                                                        const float *__restrict__ v_renders, const float *__restrict__ v_alphas,
                                                                        float *__restrict__ v_alpha_pair) {
    const uint64_t m = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    if (m >= a.M) return;
    const int64_t ray = acc_ray(a, m);
    if (m > 0 && acc_ray(a, m - 1) == ray) return;
    uint64_t end = m;
    float T = 1.f;
    for (; end < a.M && acc_ray(a, end) == ray; ++end) T *= 1.f - alpha_buf[end];
    const float va = v_alphas != nullptr ? v_alphas[ray] : 0.f;
    const float *vr = v_renders != nullptr ? v_renders + ray * a.channels : nullptr;
    float S = 0.f; // sum over the entries behind j of g_k w_k
    for (uint64_t j = end; j-- > m;) {
        const float al = alpha_buf[j];
        const float ra = 1.f / (1.f - al); // alpha <= 0.999
        T *= ra;                           // transmittance in front of j
        float g = va;                      // d L / d weight_j
        if (vr != nullptr) {
            const float *col = a.colors + ((size_t)a.cids[j] * a.N + (size_t)a.gids[j]) * a.channels;
            for (uint32_t k = 0; k < a.channels; ++k) g += vr[k] * col[k];
        }
        v_alpha_pair[j] = T * g - S * ra;
        S += g * al * T;
    }
}

__global__ void __launch_bounds__(GS_BLOCK) accumulate_bwd_scatter_kernel(AccArgs a, const float *__restrict__ weights,
                                                                          const float *__restrict__ v_alpha_pair, const float *__restrict__ v_renders,
                                                                          float *__restrict__ v_means2d, float *__restrict__ v_conics,
                                                                          float *__restrict__ v_opacities, float *__restrict__ v_colors) {
    const uint64_t m = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    if (m >= a.M) return;
    const uint32_t slice = blockIdx.y;
    if (slice > 0) { // colours: v_colors[elem, ch] += weight v_renders[ray, ch]
        if (v_colors == nullptr || v_renders == nullptr) return;
        const size_t elem = (size_t)a.cids[m] * a.N + (size_t)a.gids[m];
        const float w = weights[m];
        const float *vr = v_renders + acc_ray(a, m) * a.channels;
        const uint32_t c0 = (slice - 1u) * 4u;
#pragma unroll
        for (uint32_t k = 0; k < 4u; ++k)
            if (c0 + k < a.channels) atomicAdd(v_colors + elem * a.channels + c0 + k, w * vr[c0 + k]);
        return;
    }
    const AccAlpha r = acc_alpha(a, m);
    const float va = v_alpha_pair[m];
    if (!(r.raw <= 0.999f)) return; // clamp_max: no gradient above the cap (torch passes it at equality)
    if (v_opacities != nullptr) atomicAdd(v_opacities + r.elem, va * r.e);
    const float vs = -va * r.raw; // d / d sigma
    if (v_conics != nullptr) {
        atomicAdd(v_conics + 3 * r.elem, 0.5f * r.dx * r.dx * vs);
        atomicAdd(v_conics + 3 * r.elem + 1, r.dx * r.dy * vs);
        atomicAdd(v_conics + 3 * r.elem + 2, 0.5f * r.dy * r.dy * vs);
    }
    if (v_means2d != nullptr) { // delta = pixel - mean
        const float ca = a.conics[3 * r.elem], cb = a.conics[3 * r.elem + 1], cc = a.conics[3 * r.elem + 2];
        atomicAdd(v_means2d + 2 * r.elem, -(ca * r.dx + cb * r.dy) * vs);
        atomicAdd(v_means2d + 2 * r.elem + 1, -(cc * r.dy + cb * r.dx) * vs);
    }
}

} // namespace

extern "C" int32_t gs_accumulate_fwd(uint64_t M, uint32_t C, uint32_t N, uint32_t channels, const float *means2d, const float *conics,
                                     const float *opacities, const float *colors, const int64_t *gaussian_ids, const int64_t *pixel_ids,
                                     const int64_t *camera_ids, int32_t image_width, int32_t image_height, float *alpha_buf, float *weights,
                                     float *renders, float *alphas, gs_stream_t stream) {
    if (M == 0) return 0;
    GS_CHECK_ARG(means2d && conics && opacities && colors && gaussian_ids && pixel_ids && camera_ids && alpha_buf && weights && renders && alphas,
                 "null pointer");
    GS_CHECK_ARG(channels > 0 && image_width > 0 && image_height > 0, "channels, image_width and image_height must be > 0");
    GS_CHECK_ARG(M <= 0x7FFFFFFFull * GS_BLOCK, "too many intersections");
    const AccArgs a = {M, C, N, channels, image_width, image_height, means2d, conics, opacities, colors, gaussian_ids, pixel_ids, camera_ids};
    hipLaunchKernelGGL(accumulate_fwd_kernel, dim3(gs_div_up(M, GS_BLOCK), 1 + gs_div_up(channels, 4)), dim3(GS_BLOCK), 0, (hipStream_t)stream, a,
                       alpha_buf, weights, renders, alphas);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_accumulate_bwd(uint64_t M, uint32_t C, uint32_t N, uint32_t channels, const float *means2d, const float *conics,
                                     const float *opacities, const float *colors, const int64_t *gaussian_ids, const int64_t *pixel_ids,
                                     const int64_t *camera_ids, int32_t image_width, int32_t image_height, const float *alpha_buf,
                                     const float *weights, const float *v_renders, const float *v_alphas, float *v_alpha_pair,
                                     float *v_means2d, float *v_conics, float *v_opacities, float *v_colors, gs_stream_t stream) {
    if (M == 0) return 0;
    GS_CHECK_ARG(means2d && conics && opacities && colors && gaussian_ids && pixel_ids && camera_ids && alpha_buf && weights && v_alpha_pair,
                 "null pointer");
    GS_CHECK_ARG(channels > 0 && image_width > 0 && image_height > 0, "channels, image_width and image_height must be > 0");
    const AccArgs a = {M, C, N, channels, image_width, image_height, means2d, conics, opacities, colors, gaussian_ids, pixel_ids, camera_ids};
    hipLaunchKernelGGL(accumulate_bwd_alpha_kernel, dim3(gs_div_up(M, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream, a, alpha_buf, v_renders,
                       v_alphas, v_alpha_pair);
    hipLaunchKernelGGL(accumulate_bwd_scatter_kernel, dim3(gs_div_up(M, GS_BLOCK), 1 + gs_div_up(channels, 4)), dim3(GS_BLOCK), 0,
                       (hipStream_t)stream, a, weights, v_alpha_pair, v_renders, v_means2d, v_conics, v_opacities, v_colors);
    GS_CHECK_LAUNCH();
    return 0;
}
