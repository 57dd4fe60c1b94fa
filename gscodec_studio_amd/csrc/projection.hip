// projection.hip -- R1: fully fused 3D->2D projection of gaussians, fwd + bwd (gfx950).
//
// Replaces the reference kernels
//   gsplat/cuda/csrc/fully_fused_projection_fwd.cu:22-196
//   gsplat/cuda/csrc/fully_fused_projection_bwd.cu:24-263
//   gsplat/cuda/csrc/fully_fused_projection_packed_{fwd,bwd}.cu
// Design notes (MI355X):
//   * HBM-streaming kernels: one lane per (camera, gaussian) in fwd, one lane per
//     gaussian looping over cameras in bwd.  The bwd therefore needs NO atomics and
//     no warp-segmented reduction (the reference uses cg::labeled_partition, which
//     does not exist on ROCm) and is deterministic; every gaussian row of the gradient
//     tensors is written exactly once, so the caller does not zero-fill them.
//   * Camera constants (viewmat / K, 100 B per camera) are wave-uniform loads and live
//     in SGPRs.
//   * All maths are written on symmetric 2x2 / 3x3 forms (6 / 3 scalars), row-major.
#include "gs_common.h"
#include "proj_models.h"
#include "projection_dev.h"
#include "sh_eval.h"
#include "sh_bwd_lane.h"

namespace {


// ROWS: the outputs are splat rows (include/gsplat_hip.h): one 64-byte row per (camera, gaussian) carrying everything the
// compositing kernels fetch -- mean2d, conic, opacity (x antialias compensation), optionally the colour -- plus depth,
// radius and compensation; `means2d` is the row buffer, `conics` is unused.  radii / depths are ALSO written densely (what
// the binning kernels stream through).
struct RowExtras {
    const float *opacities; // [N] or NULL
    const float *colors;    // [N,3] post-activation colours or NULL
    int antialiased;        // opacity column = opacity * compensation
    // SH colours evaluated in the same pass (what gs_sh_view_fwd would write into the rows afterwards): the visible splat's
    // coefficient row is read here, the whole 48 bytes of the row leave in three full 16-byte stores, and the colour launch
    // with its second pass over means / radii disappears
    const float *sh_coeffs; // [N,K,3] or NULL
    const float *sh_rest;   // split rows (sh_eval.h): sh_coeffs is [N,1,3], this [N,K-1,3]
    uint32_t sh_K, sh_degree;
    int sh_vec;             // coefficient rows are 16-byte aligned (dwordx4 loads)
    // the shN mask of the compression-simulation hooks applied on the fly (split rows only): coefficients of the bands >= 1
    // times gs_mask_value(logit[n]) -- the masked coefficients are never materialised
    const float *sh_mask_logits; // [N] or NULL
    float sh_mask_temp;
    int sh_mask_binary;
    // the binning's intersection count in the same pass (what gs_isect_count_keys would compute from the rows afterwards): the
    // host learns n_isects one kernel after the step starts, with the whole depth pre-sort still queued behind it
    int32_t *tiles_per_gauss; // [C,N] or NULL
    int32_t *block_sums;      // [C * gridDim.x][2] (pinned host memory) or NULL: (intersections, visible pairs) per workgroup
    float tile_size;
    int32_t tile_width, tile_height;
};

// SHMODE: -1 = no SH colours; else 3 * degree + kind, kind 0 = coefficient rows read with scalar loads, 1 = 16-byte aligned
// rows (dwordx4 loads), 2 = split rows (sh0 | shN, sh_eval.h).  A template parameter, not a switch inside the kernel: the
// register allocation of ONE kernel holding all five degrees is that of degree 4 (161 VGPRs, 3 waves per SIMD; the degree-3
// instance needs ~100).
template <bool ROWS, int SHMODE>
__global__ void __launch_bounds__(GS_BLOCK) projection_fwd_kernel(
    uint32_t C, uint32_t N,
    const float *__restrict__ means, const float *__restrict__ covars,
    const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ viewmats, const float *__restrict__ Ks,
    int W, int H, float eps2d, float near_plane, float far_plane, float radius_clip,
    int camera_model,
    int32_t *__restrict__ radii, float *__restrict__ means2d, float *__restrict__ depths,
    float *__restrict__ conics, float *__restrict__ compensations, RowExtras rx) {
    // grid = (ceil(N/256), C): the camera index is block-uniform => camera constants in SGPRs
    uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    uint32_t c = blockIdx.y;
    if (!(ROWS && rx.tiles_per_gauss != nullptr) && n >= N) return;
    const bool in = n < N;
    Camera cam = load_camera(viewmats, Ks, c);
    Splat2D s;
    s.radius = 0;
    if (in) s = project_one<false>(cam, means, covars, quats, scales, n, W, H, eps2d, near_plane, far_plane, radius_clip, camera_model);
    size_t idx = (size_t)c * N + n;
    if (ROWS && rx.tiles_per_gauss != nullptr) { // (uniform) every thread of the workgroup takes part in the sum
        rows_count_tiles(s, in, idx, rx.tiles_per_gauss, rx.block_sums, rx.tile_size, rx.tile_width, rx.tile_height);
        if (!in) return;
    }
    radii[idx] = s.radius;
    if (s.radius <= 0) return;
    if (ROWS) {
        float *row = means2d + GS_ROW_FLOATS * idx;
        depths[idx] = s.depth;
        float op = rx.opacities != nullptr ? rx.opacities[n] : 0.f;
        if (rx.antialiased) op *= s.comp;
        reinterpret_cast<float4 *>(row)[0] = make_float4(s.mx, s.my, s.ca, s.cb);
        if (rx.colors != nullptr || rx.sh_coeffs != nullptr) {
            float c0, c1, c2;
            if (SHMODE >= 0) {
                // view direction = mean - camera centre (the centre from the view matrix, wave-uniform), colour =
                // clamp_min(SH + 0.5, 0): gsplat/rendering.py:368-392 of the reference
                float cx, cy, cz;
                camera_center(viewmats + 16 * c, cx, cy, cz);
                const float *p = means + 3 * (size_t)n;
                const float dx = p[0] - cx, dy = p[1] - cy, dz = p[2] - cz;
                const float *crow = rx.sh_coeffs + (size_t)n * (rx.sh_rest != nullptr ? 3u : rx.sh_K * 3);
                const float *rest = rx.sh_rest != nullptr ? rx.sh_rest + (size_t)n * (rx.sh_K - 1) * 3 : nullptr;
                float bm = 1.f;
                const float *bmp = nullptr;
                if ((SHMODE % 3) == 2 && rx.sh_mask_logits != nullptr) { // (uniform)
                    bm = gs_mask_value(rx.sh_mask_logits[n], rx.sh_mask_temp, rx.sh_mask_binary);
                    bmp = &bm;
                }
                if constexpr (SHMODE >= 0) sh_view_color<SHMODE / 3, (SHMODE % 3) == 1, (SHMODE % 3) == 2 ? 1 : 0>(dx, dy, dz, crow, rest, true, c0, c1, c2, bmp);
                else c0 = c1 = c2 = 0.f;
            } else {
                const float *cp = rx.colors + 3 * (size_t)n;
                c0 = cp[0]; c1 = cp[1]; c2 = cp[2];
            }
            reinterpret_cast<float4 *>(row)[1] = make_float4(s.cc, op, c0, c1);
            reinterpret_cast<float4 *>(row)[2] = make_float4(c2, s.depth, __int_as_float(s.radius), s.comp);
        } else { // the colour columns belong to the SH kernel (gs_sh_view_fwd writes them into the same rows)
            reinterpret_cast<float2 *>(row)[2] = make_float2(s.cc, op);
            row[GS_ROW_DEPTH] = s.depth;
            reinterpret_cast<float2 *>(row)[5] = make_float2(__int_as_float(s.radius), s.comp);
        }
        return;
    }
    means2d[2 * idx] = s.mx;
    means2d[2 * idx + 1] = s.my;
    depths[idx] = s.depth;
    conics[3 * idx] = s.ca;
    conics[3 * idx + 1] = s.cb;
    conics[3 * idx + 2] = s.cc;
    if (compensations != nullptr) compensations[idx] = s.comp;
}

// ---------------------------------------------------------------------------
// backward
// ---------------------------------------------------------------------------

// write the per-gaussian gradient rows
GS_DEV void store_gaussian_grads(
    const ProjGrad &g, uint32_t n, bool any,
    const float *__restrict__ covars, const float *__restrict__ quats,
    const float *__restrict__ scales,
    float *__restrict__ v_means, float *__restrict__ v_covars,
    float *__restrict__ v_quats, float *__restrict__ v_scales, const float *__restrict__ v_means_add) {
    if (v_means != nullptr) {
        // v_means_add: a contribution to d/d means that reached the caller by another path (the view directions of the SH
        // colours); summed here instead of by a separate elementwise pass over [N,3]
        float ax = 0.f, ay = 0.f, az = 0.f;
        if (v_means_add != nullptr) {
            ax = v_means_add[3 * (size_t)n]; ay = v_means_add[3 * (size_t)n + 1]; az = v_means_add[3 * (size_t)n + 2];
        }
        v_means[3 * (size_t)n] = g.v_px + ax;
        v_means[3 * (size_t)n + 1] = g.v_py + ay;
        v_means[3 * (size_t)n + 2] = g.v_pz + az;
    }
    if (covars != nullptr) {
        if (v_covars != nullptr) {
            float *o = v_covars + 6 * (size_t)n;
            // off-diagonals carry both (i,j) and (j,i): fully_fused_projection_bwd.cu:217-222
            o[0] = g.v_S.xx; o[1] = 2.f * g.v_S.xy; o[2] = 2.f * g.v_S.xz;
            o[3] = g.v_S.yy; o[4] = 2.f * g.v_S.yz; o[5] = g.v_S.zz;
        }
    } else {
        float vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
        if (any) {
            const float *q = quats + 4 * (size_t)n;
            const float *s = scales + 3 * (size_t)n;
            Mat3 R = quat_to_rotmat(q[0], q[1], q[2], q[3]);
            covar_vjp_quat_scale(q[0], q[1], q[2], q[3], s[0], s[1], s[2], R, sym3_to_mat3(g.v_S), vq, vs);
        }
        if (v_quats != nullptr) {
            float *o = v_quats + 4 * (size_t)n;
            o[0] = vq[0]; o[1] = vq[1]; o[2] = vq[2]; o[3] = vq[3];
        }
        if (v_scales != nullptr) {
            float *o = v_scales + 3 * (size_t)n;
            o[0] = vs[0]; o[1] = vs[1]; o[2] = vs[2];
        }
    }
}

// Row form of the backward (rg.rows != NULL): conics / compensations come from the splat rows, the gradients of mean2d /
// conic / opacity / colour from the compositing backward's gradient rows (same columns); this kernel then also sums the
// opacity and colour gradients over the cameras (what autograd's `opacities.repeat(C, 1)` / `colors.expand` backward and
// the antialias multiply did in passes of their own).
struct RowGrads {
    const float *rows;      // [C,N,16] or NULL (array form)
    const float *grad_rows; // [C,N,16]
    const float *opacities; // [N] (antialiased only)
    float *v_opacities;     // [N] or NULL
    float *v_colors;        // [N,3] or NULL
    int antialiased;
    int prefilled;          // every per-gaussian output holds zeros already: gaussians no camera sees are not stored
    // SH colours evaluated by the forward (gs_projection_rows_fwd with sh_coeffs): their backward runs in this pass too
    const float *sh_coeffs; // [N,K,3], or the DC band [N,1,3] with sh_rest
    const float *sh_rest;   // [N,K-1,3] or NULL
    uint32_t sh_K;
    float *v_sh;            // gradient of sh_coeffs
    float *v_sh_rest;       // gradient of sh_rest
    const float *sh_mask_logits; // the forward's shN mask (split rows), or NULL
    float sh_mask_temp;
    int sh_mask_binary;
    float *v_sh_mask_logits;     // [N] gradient of the logits (every entry written), or NULL
};

// SHDEG >= 0 (row form, shared coefficients, fixed poses): the SH backward of the colours the forward evaluated runs FIRST in
// the same lane (sh_bwd_lane.h: v_sh rows out, d/d view direction kept in three registers), then the projection chain --
// one pass over radii / means / the two row buffers instead of two, and no [N,3] round trip for the direction gradient
// (sh_bwd_kernel 40.4 us + projection_bwd_kernel 20.4 us -> 52.3 us at BASELINE config 2).
template <bool NEED_VIEW, int SHDEG = -1>
__global__ void __launch_bounds__(GS_BLOCK) projection_bwd_kernel(
    uint32_t C, uint32_t N,
    const float *__restrict__ means, const float *__restrict__ covars,
    const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ viewmats, const float *__restrict__ Ks,
    int W, int H, float eps2d, int camera_model,
    const int32_t *__restrict__ radii, const float *__restrict__ conics,
    const float *__restrict__ compensations,
    const float *__restrict__ v_means2d, const float *__restrict__ v_depths,
    const float *__restrict__ v_conics, const float *__restrict__ v_compensations,
    float *__restrict__ v_means, float *__restrict__ v_covars, float *__restrict__ v_quats,
    float *__restrict__ v_scales, float *__restrict__ v_viewmats, uint32_t s_m2, uint32_t s_cn,
    const float *__restrict__ v_means_add, RowGrads rg) {
    __shared__ float s_view[GS_BLOCK / GS_WAVE][12];
    float v_op = 0.f, v_c0 = 0.f, v_c1 = 0.f, v_c2 = 0.f;
    uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    bool in_range = n < N;
    float shx = 0.f, shy = 0.f, shz = 0.f; // d/d means through the SH view directions
    if (SHDEG >= 0) {
        const ShView view = {means, viewmats, radii, 1, 1, nullptr, nullptr, nullptr, 0u, nullptr, rg.sh_rest, rg.v_sh_rest,
                             (uint32_t)GS_ROW_FLOATS, rg.prefilled, rg.sh_mask_logits, rg.sh_mask_temp, rg.sh_mask_binary,
                             rg.v_sh_mask_logits};
        bool any_sh;
        sh_bwd_lane<(SHDEG >= 0 ? SHDEG : 0), true, true>(C, N, rg.sh_K, n, in_range, nullptr, rg.sh_coeffs, nullptr, rg.grad_rows + GS_ROW_COLOR,
                                                          rg.v_sh, nullptr, view, rg.rows + GS_ROW_COLOR, (uint32_t)GS_ROW_FLOATS,
                                                          v_means != nullptr, shx, shy, shz, any_sh);
    }
    float px = 0.f, py = 0.f, pz = 0.f;
    Sym3 S = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    bool loaded = false;
    ProjGrad g;
    grad_zero(g);
    bool any = false;
    for (uint32_t c = 0; c < C; ++c) {
        size_t idx = (size_t)c * N + n;
        bool vis = in_range && radii[idx] > 0;
        if (NEED_VIEW) {
            g.v_W = mat3_zero();
            g.v_t[0] = g.v_t[1] = g.v_t[2] = 0.f;
        }
        if (vis) {
            if (!loaded) {
                const float *p = means + 3 * (size_t)n;
                px = p[0]; py = p[1]; pz = p[2];
                S = load_covar(covars, quats, scales, n);
                loaded = true;
            }
            Camera cam = load_camera(viewmats, Ks, c);
            if (rg.rows != nullptr) {
                const float4 *r = reinterpret_cast<const float4 *>(rg.rows + GS_ROW_FLOATS * idx);
                const float4 *gr = reinterpret_cast<const float4 *>(rg.grad_rows + GS_ROW_FLOATS * idx);
                const float4 r0 = r[0], g0 = gr[0], g1 = gr[1];
                const float cc = reinterpret_cast<const float *>(r)[GS_ROW_CONIC + 2];
                const float comp = rg.antialiased ? reinterpret_cast<const float *>(r)[GS_ROW_COMPENSATION] : 1.f;
                const float v_opac_cn = g1.y;
                const float v_comp = rg.antialiased ? v_opac_cn * rg.opacities[n] : 0.f;
                v_op += v_opac_cn * comp;
                v_c0 += g1.z;
                v_c1 += g1.w;
                if (rg.v_colors != nullptr) v_c2 += reinterpret_cast<const float *>(gr)[GS_ROW_COLOR + 2];
                project_one_vjp<NEED_VIEW>(cam, px, py, pz, S, W, H, eps2d, camera_model, r0.z, r0.w, cc, comp, v_comp,
                                           rg.antialiased != 0, g0.x, g0.y, v_depths != nullptr ? v_depths[idx] : 0.f,
                                           g0.z, g0.w, g1.x, g);
            } else {
            bool has_comp = v_compensations != nullptr;
            project_one_vjp<NEED_VIEW>(
                cam, px, py, pz, S, W, H, eps2d, camera_model,
                conics[3 * idx], conics[3 * idx + 1], conics[3 * idx + 2],
                has_comp ? compensations[idx] : 0.f, has_comp ? v_compensations[idx] : 0.f, has_comp,
                v_means2d[s_m2 * idx], v_means2d[s_m2 * idx + 1], v_depths != nullptr ? v_depths[idx] : 0.f,
                v_conics[s_cn * idx], v_conics[s_cn * idx + 1], v_conics[s_cn * idx + 2], g);
            }
            any = true;
        }
        if (NEED_VIEW) {
            // block reduce the 12 viewmat entries for camera c, one atomic set per block
            float vals[12];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
#pragma unroll
                for (int j = 0; j < 3; ++j) vals[4 * i + j] = g.v_W.m[i][j];
                vals[4 * i + 3] = g.v_t[i];
            }
            uint32_t wave = threadIdx.x / GS_WAVE, lane = threadIdx.x % GS_WAVE;
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                float s = wave_sum(vals[k]);
                if (lane == 0) s_view[wave][k] = s;
            }
            __syncthreads();
            if (threadIdx.x < 12) {
                float s = 0.f;
#pragma unroll
                for (int w = 0; w < GS_BLOCK / GS_WAVE; ++w) s += s_view[w][threadIdx.x];
                if (s != 0.f) atomicAdd(v_viewmats + 16 * c + threadIdx.x, s);
            }
            __syncthreads();
        }
    }
    if (in_range && (any || !rg.prefilled)) {
        if (SHDEG >= 0) { g.v_px += shx; g.v_py += shy; g.v_pz += shz; }
        store_gaussian_grads(g, n, any, covars, quats, scales, v_means, v_covars, v_quats, v_scales, v_means_add);
        if (rg.v_opacities != nullptr) rg.v_opacities[n] = v_op;
        if (rg.v_colors != nullptr) {
            rg.v_colors[3 * (size_t)n] = v_c0;
            rg.v_colors[3 * (size_t)n + 1] = v_c1;
            rg.v_colors[3 * (size_t)n + 2] = v_c2;
        }
    }
}

// ---------------------------------------------------------------------------
// packed (COO) variants
// ---------------------------------------------------------------------------
template <bool FILL>
__global__ void __launch_bounds__(GS_BLOCK) projection_packed_kernel(
    uint32_t C, uint32_t N,
    const float *__restrict__ means, const float *__restrict__ covars,
    const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ viewmats, const float *__restrict__ Ks,
    int W, int H, float eps2d, float near_plane, float far_plane, float radius_clip,
    int camera_model,
    const int32_t *__restrict__ block_accum, int32_t *__restrict__ block_cnts,
    int32_t *__restrict__ indptr, int64_t *__restrict__ camera_ids,
    int64_t *__restrict__ gaussian_ids, int32_t *__restrict__ radii,
    float *__restrict__ means2d, float *__restrict__ depths, float *__restrict__ conics,
    float *__restrict__ compensations) {
    __shared__ int32_t s_wave_cnt[GS_BLOCK / GS_WAVE];
    uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    uint32_t c = blockIdx.y;
    uint32_t nblocks = gridDim.x;
    Splat2D s;
    s.radius = 0;
    if (n < N) {
        Camera cam = load_camera(viewmats, Ks, c);
        s = project_one<true>(cam, means, covars, quats, scales, n, W, H, eps2d, near_plane, far_plane,
                              radius_clip, camera_model);
    }
    bool vis = s.radius > 0;
    unsigned long long ballot = __ballot(vis);
    uint32_t wave = threadIdx.x / GS_WAVE, lane = threadIdx.x % GS_WAVE;
    if (lane == 0) s_wave_cnt[wave] = __popcll(ballot);
    __syncthreads();
    uint32_t block_id = c * nblocks + blockIdx.x;
    if (!FILL) {
        if (threadIdx.x == 0) {
            int32_t t = 0;
#pragma unroll
            for (int w = 0; w < GS_BLOCK / GS_WAVE; ++w) t += s_wave_cnt[w];
            block_cnts[block_id] = t;
        }
        return;
    }
    int32_t base = block_id == 0 ? 0 : block_accum[block_id - 1];
    for (uint32_t w = 0; w < wave; ++w) base += s_wave_cnt[w];
    if (vis) {
        unsigned long long lt = (lane == 0) ? 0ull : (ballot & ((1ull << lane) - 1ull));
        size_t o = (size_t)base + __popcll(lt);
        camera_ids[o] = c;
        gaussian_ids[o] = n;
        radii[o] = s.radius;
        means2d[2 * o] = s.mx;
        means2d[2 * o + 1] = s.my;
        depths[o] = s.depth;
        conics[3 * o] = s.ca;
        conics[3 * o + 1] = s.cb;
        conics[3 * o + 2] = s.cc;
        if (compensations != nullptr) compensations[o] = s.comp;
    }
    // CSR row pointer: first block of every camera row
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        indptr[c] = block_id == 0 ? 0 : block_accum[block_id - 1];
        if (c == C - 1) indptr[C] = block_accum[C * nblocks - 1];
    }
}

template <bool NEED_VIEW>
__global__ void __launch_bounds__(GS_BLOCK) projection_packed_bwd_kernel(
    uint32_t C, uint32_t N, uint32_t nnz,
    const float *__restrict__ means, const float *__restrict__ covars,
    const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ viewmats, const float *__restrict__ Ks,
    int W, int H, float eps2d, int camera_model,
    const int64_t *__restrict__ camera_ids, const int64_t *__restrict__ gaussian_ids,
    const float *__restrict__ conics, const float *__restrict__ compensations,
    const float *__restrict__ v_means2d, const float *__restrict__ v_depths,
    const float *__restrict__ v_conics, const float *__restrict__ v_compensations,
    int sparse_grad,
    float *__restrict__ v_means, float *__restrict__ v_covars, float *__restrict__ v_quats,
    float *__restrict__ v_scales, float *__restrict__ v_viewmats) {
    uint32_t idx = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (idx >= nnz) return;
    uint32_t c = (uint32_t)camera_ids[idx];
    uint32_t n = (uint32_t)gaussian_ids[idx];
    const float *p = means + 3 * (size_t)n;
    float px = p[0], py = p[1], pz = p[2];
    Sym3 S = load_covar(covars, quats, scales, n);
    Camera cam = load_camera(viewmats, Ks, c);
    ProjGrad g;
    grad_zero(g);
    bool has_comp = v_compensations != nullptr;
    project_one_vjp<NEED_VIEW>(
        cam, px, py, pz, S, W, H, eps2d, camera_model, conics[3 * (size_t)idx],
        conics[3 * (size_t)idx + 1], conics[3 * (size_t)idx + 2],
        has_comp ? compensations[idx] : 0.f, has_comp ? v_compensations[idx] : 0.f, has_comp,
        v_means2d[2 * (size_t)idx], v_means2d[2 * (size_t)idx + 1], v_depths != nullptr ? v_depths[idx] : 0.f,
        v_conics[3 * (size_t)idx], v_conics[3 * (size_t)idx + 1], v_conics[3 * (size_t)idx + 2], g);

    float vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f};
    if (covars == nullptr) {
        const float *q = quats + 4 * (size_t)n;
        const float *s = scales + 3 * (size_t)n;
        Mat3 R = quat_to_rotmat(q[0], q[1], q[2], q[3]);
        covar_vjp_quat_scale(q[0], q[1], q[2], q[3], s[0], s[1], s[2], R, sym3_to_mat3(g.v_S), vq, vs);
    }
    if (sparse_grad) {
        size_t o = idx;
        if (v_means != nullptr) {
            v_means[3 * o] = g.v_px; v_means[3 * o + 1] = g.v_py; v_means[3 * o + 2] = g.v_pz;
        }
        if (covars != nullptr) {
            if (v_covars != nullptr) {
                float *d = v_covars + 6 * o;
                d[0] = g.v_S.xx; d[1] = 2.f * g.v_S.xy; d[2] = 2.f * g.v_S.xz;
                d[3] = g.v_S.yy; d[4] = 2.f * g.v_S.yz; d[5] = g.v_S.zz;
            }
        } else {
            if (v_quats != nullptr) {
                float *d = v_quats + 4 * o;
                d[0] = vq[0]; d[1] = vq[1]; d[2] = vq[2]; d[3] = vq[3];
            }
            if (v_scales != nullptr) {
                float *d = v_scales + 3 * o;
                d[0] = vs[0]; d[1] = vs[1]; d[2] = vs[2];
            }
        }
    } else {
        // dense: several cameras may hit the same gaussian row -> float atomics; with ONE camera every gaussian appears at
        // most once and the (zero-filled) row is simply written
        size_t o = n;
        const bool one = C == 1u;
        auto put = [&](float *dst, float v) {
            if (one) *dst = v;
            else atomicAdd(dst, v);
        };
        if (v_means != nullptr) {
            put(v_means + 3 * o, g.v_px);
            put(v_means + 3 * o + 1, g.v_py);
            put(v_means + 3 * o + 2, g.v_pz);
        }
        if (covars != nullptr) {
            if (v_covars != nullptr) {
                float *d = v_covars + 6 * o;
                put(d, g.v_S.xx); put(d + 1, 2.f * g.v_S.xy); put(d + 2, 2.f * g.v_S.xz);
                put(d + 3, g.v_S.yy); put(d + 4, 2.f * g.v_S.yz); put(d + 5, g.v_S.zz);
            }
        } else {
            if (v_quats != nullptr) {
                float *d = v_quats + 4 * o;
                put(d, vq[0]); put(d + 1, vq[1]); put(d + 2, vq[2]); put(d + 3, vq[3]);
            }
            if (v_scales != nullptr) {
                float *d = v_scales + 3 * o;
                put(d, vs[0]); put(d + 1, vs[1]); put(d + 2, vs[2]);
            }
        }
    }
    if (NEED_VIEW) {
        // rows are sorted by camera, so most waves are camera-uniform: reduce when they are
        float vals[12];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
#pragma unroll
            for (int j = 0; j < 3; ++j) vals[4 * i + j] = g.v_W.m[i][j];
            vals[4 * i + 3] = g.v_t[i];
        }
        uint32_t c0 = __builtin_amdgcn_readfirstlane(c);
        bool uniform = __all(c == c0) && (__popcll(__ballot(1)) == GS_WAVE);
        if (uniform) {
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                float s = wave_sum(vals[k]);
                if (lane_id() == 0) atomicAdd(v_viewmats + 16 * c0 + k, s);
            }
        } else {
#pragma unroll
            for (int k = 0; k < 12; ++k) atomicAdd(v_viewmats + 16 * c + k, vals[k]);
        }
    }
}

} // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
extern "C" int32_t gs_projection_fwd(
    uint32_t C, uint32_t N, const float *means, const float *covars, const float *quats,
    const float *scales, const float *viewmats, const float *Ks, int32_t image_width,
    int32_t image_height, float eps2d, float near_plane, float far_plane, float radius_clip,
    int32_t camera_model, int32_t *radii, float *means2d, float *depths, float *conics,
    float *compensations, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(means && viewmats && Ks && radii && means2d && depths && conics, "null pointer");
    GS_CHECK_ARG((covars != nullptr) != (quats != nullptr && scales != nullptr),
                 "exactly one of covars / (quats, scales) must be given");
    GS_CHECK_ARG(camera_model >= 0 && camera_model <= 2, "bad camera_model");
    dim3 grid(gs_div_up(N, GS_BLOCK), C);
    const RowExtras none = {nullptr, nullptr, 0, nullptr, nullptr, 0u, 0u, 0};
    hipLaunchKernelGGL((projection_fwd_kernel<false, -1>), grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N, means,
                       covars, quats, scales, viewmats, Ks, image_width, image_height, eps2d, near_plane,
                       far_plane, radius_clip, camera_model, radii, means2d, depths, conics, compensations, none);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" uint32_t gs_projection_rows_blocks(uint32_t N) { return gs_div_up(N, GS_BLOCK); }

extern "C" int32_t gs_projection_rows_fwd(
    uint32_t C, uint32_t N, const float *means, const float *covars, const float *quats,
    const float *scales, const float *viewmats, const float *Ks, int32_t image_width,
    int32_t image_height, float eps2d, float near_plane, float far_plane, float radius_clip,
    int32_t camera_model, const float *opacities, const float *colors, int32_t antialiased, const float *sh_coeffs, const float *sh_coeffs_rest,
    uint32_t sh_K, uint32_t sh_degree, const float *sh_mask_logits, float sh_mask_temperature, int32_t sh_mask_binary,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, int32_t *tiles_per_gauss, int32_t *block_sums,
    int32_t *radii, float *depths, float *rows, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(means && viewmats && Ks && radii && depths && rows, "null pointer");
    GS_CHECK_ARG((uintptr_t)rows % 64 == 0, "the row buffer must be 64-byte aligned");
    GS_CHECK_ARG((covars != nullptr) != (quats != nullptr && scales != nullptr),
                 "exactly one of covars / (quats, scales) must be given");
    GS_CHECK_ARG(camera_model >= 0 && camera_model <= 2, "bad camera_model");
    GS_CHECK_ARG(!antialiased || opacities != nullptr, "antialiased needs the opacities");
    GS_CHECK_ARG(sh_coeffs == nullptr || colors == nullptr, "colors and sh_coeffs exclude each other");
    GS_CHECK_ARG(sh_coeffs == nullptr || (sh_degree <= 4 && (sh_degree + 1) * (sh_degree + 1) <= sh_K), "bad SH degree / K");
    dim3 grid(gs_div_up(N, GS_BLOCK), C);
    GS_CHECK_ARG(sh_coeffs_rest == nullptr || (sh_coeffs != nullptr && sh_K >= 2), "sh_coeffs_rest needs sh_coeffs and K >= 2");
    const int sh_vec = sh_coeffs != nullptr && (sh_coeffs_rest != nullptr || (uintptr_t)sh_coeffs % 16 == 0) && ((sh_K * 3u) % 4u == 0);
    GS_CHECK_ARG(sh_mask_logits == nullptr || sh_coeffs_rest != nullptr, "the shN mask needs split coefficient rows (sh_coeffs_rest)");
    GS_CHECK_ARG(sh_mask_logits == nullptr || sh_mask_binary || sh_mask_temperature > 0.f, "the mask temperature must be positive");
    GS_CHECK_ARG(tiles_per_gauss != nullptr || block_sums == nullptr, "block_sums come with tiles_per_gauss");
    GS_CHECK_ARG(tiles_per_gauss == nullptr || tile_size > 0, "tile_size must be > 0");
    const RowExtras rx = {opacities, colors, antialiased, sh_coeffs, sh_coeffs_rest, sh_K, sh_degree, sh_vec, sh_mask_logits, sh_mask_temperature,
                          sh_mask_binary, tiles_per_gauss, block_sums, (float)tile_size, (int32_t)tile_width, (int32_t)tile_height};
    const int shmode = sh_coeffs == nullptr ? -1 : (int)sh_degree * 3 + (sh_coeffs_rest != nullptr ? 2 : (sh_vec ? 1 : 0));
#define GS_ROWS_LAUNCH(M)                                                                                                        \
    case M:                                                                                                                      \
        hipLaunchKernelGGL((projection_fwd_kernel<true, M>), grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N, means, covars,   \
                           quats, scales, viewmats, Ks, image_width, image_height, eps2d, near_plane, far_plane, radius_clip,    \
                           camera_model, radii, rows, depths, (float *)nullptr, (float *)nullptr, rx);                           \
        break;
    switch (shmode) {
        GS_ROWS_LAUNCH(-1) GS_ROWS_LAUNCH(0) GS_ROWS_LAUNCH(1) GS_ROWS_LAUNCH(2) GS_ROWS_LAUNCH(3) GS_ROWS_LAUNCH(4) GS_ROWS_LAUNCH(5)
        GS_ROWS_LAUNCH(6) GS_ROWS_LAUNCH(7) GS_ROWS_LAUNCH(8) GS_ROWS_LAUNCH(9) GS_ROWS_LAUNCH(10) GS_ROWS_LAUNCH(11)
        GS_ROWS_LAUNCH(12) GS_ROWS_LAUNCH(13) GS_ROWS_LAUNCH(14)
    }
#undef GS_ROWS_LAUNCH
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_projection_rows_bwd(
    uint32_t C, uint32_t N, const float *means, const float *covars, const float *quats,
    const float *scales, const float *viewmats, const float *Ks, int32_t image_width,
    int32_t image_height, float eps2d, int32_t camera_model, const int32_t *radii, const float *rows,
    const float *grad_rows, const float *v_depths, const float *opacities, int32_t antialiased, float *v_means,
    float *v_covars, float *v_quats, float *v_scales, float *v_viewmats, float *v_opacities, float *v_colors,
    const float *v_means_add, const float *sh_coeffs, const float *sh_coeffs_rest, uint32_t sh_K, uint32_t sh_degree,
    float *v_sh_coeffs, float *v_sh_coeffs_rest, const float *sh_mask_logits, float sh_mask_temperature, int32_t sh_mask_binary,
    float *v_sh_mask_logits, int32_t outputs_prefilled, gs_stream_t stream) {
    if (N == 0) return 0;
    GS_CHECK_ARG(means && viewmats && Ks && radii && rows && grad_rows, "null pointer");
    GS_CHECK_ARG((uintptr_t)rows % 16 == 0 && (uintptr_t)grad_rows % 16 == 0, "row buffers must be 16-byte aligned");
    GS_CHECK_ARG((covars != nullptr) != (quats != nullptr && scales != nullptr),
                 "exactly one of covars / (quats, scales) must be given");
    GS_CHECK_ARG(!antialiased || opacities != nullptr, "antialiased needs the opacities");
    dim3 grid(gs_div_up(N, GS_BLOCK));
    RowGrads rg = {rows, grad_rows, opacities, v_opacities, v_colors, antialiased, outputs_prefilled != 0, nullptr, nullptr, 0u, nullptr, nullptr,
                   nullptr, 1.f, 0, nullptr};
    GS_CHECK_ARG(sh_mask_logits == nullptr || (sh_coeffs != nullptr && sh_coeffs_rest != nullptr), "the shN mask needs the fused SH backward with split rows");
    GS_CHECK_ARG(v_sh_mask_logits == nullptr || (sh_mask_logits != nullptr && !sh_mask_binary), "v_sh_mask_logits needs the (training-mode) mask");
    const float *nul = nullptr;
    if (sh_coeffs != nullptr) {
        // the SH backward in the same pass: the vectorised row form only (what gs_sh_view_bwd stages through LDS), fixed poses
        GS_CHECK_ARG(v_viewmats == nullptr && v_colors == nullptr, "fused SH backward: no camera-pose / per-gaussian colour gradients");
        GS_CHECK_ARG(v_sh_coeffs != nullptr && sh_degree <= 4 && (sh_degree + 1) * (sh_degree + 1) <= sh_K, "fused SH backward: v_sh_coeffs, degree <= 4, K >= (degree + 1)^2");
        GS_CHECK_ARG((sh_coeffs_rest == nullptr) == (v_sh_coeffs_rest == nullptr), "sh_coeffs_rest and v_sh_coeffs_rest go together");
        GS_CHECK_ARG((sh_K * 3u) % 4u == 0 && (uintptr_t)v_sh_coeffs % 16 == 0 && (uintptr_t)v_sh_coeffs_rest % 16 == 0 &&
                         (sh_coeffs_rest != nullptr ? sh_K >= 2 : (uintptr_t)sh_coeffs % 16 == 0),
                     "fused SH backward needs 3 K % 4 == 0 and 16-byte aligned rows (gs_projection_rows_bwd_sh_ok); call gs_sh_view_bwd instead");
        rg.sh_coeffs = sh_coeffs; rg.sh_rest = sh_coeffs_rest; rg.sh_K = sh_K; rg.v_sh = v_sh_coeffs; rg.v_sh_rest = v_sh_coeffs_rest;
        rg.sh_mask_logits = sh_mask_logits; rg.sh_mask_temp = sh_mask_temperature; rg.sh_mask_binary = sh_mask_binary;
        rg.v_sh_mask_logits = v_sh_mask_logits;
#define GS_ROWS_BWD_SH(D)                                                                                                        \
    case D:                                                                                                                      \
        hipLaunchKernelGGL((projection_bwd_kernel<false, D>), grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N, means, covars,  \
                           quats, scales, viewmats, Ks, image_width, image_height, eps2d, camera_model, radii, nul, nul, nul,     \
                           v_depths, nul, nul, v_means, v_covars, v_quats, v_scales, v_viewmats, 16u, 16u, v_means_add, rg);      \
        break;
        switch (sh_degree) { GS_ROWS_BWD_SH(0) GS_ROWS_BWD_SH(1) GS_ROWS_BWD_SH(2) GS_ROWS_BWD_SH(3) GS_ROWS_BWD_SH(4) }
#undef GS_ROWS_BWD_SH
    } else if (v_viewmats != nullptr) {
        hipLaunchKernelGGL(projection_bwd_kernel<true>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N,
                           means, covars, quats, scales, viewmats, Ks, image_width, image_height, eps2d,
                           camera_model, radii, nul, nul, nul, v_depths, nul, nul, v_means, v_covars, v_quats, v_scales,
                           v_viewmats, 16u, 16u, v_means_add, rg);
    } else {
        hipLaunchKernelGGL(projection_bwd_kernel<false>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N,
                           means, covars, quats, scales, viewmats, Ks, image_width, image_height, eps2d,
                           camera_model, radii, nul, nul, nul, v_depths, nul, nul, v_means, v_covars, v_quats, v_scales,
                           v_viewmats, 16u, 16u, v_means_add, rg);
    }
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_projection_bwd(
    uint32_t C, uint32_t N, const float *means, const float *covars, const float *quats,
    const float *scales, const float *viewmats, const float *Ks, int32_t image_width,
    int32_t image_height, float eps2d, int32_t camera_model, const int32_t *radii,
    const float *conics, const float *compensations, const float *v_means2d,
    const float *v_depths, const float *v_conics, const float *v_compensations, float *v_means,
    float *v_covars, float *v_quats, float *v_scales, float *v_viewmats, uint32_t v_means2d_stride,
    uint32_t v_conics_stride, const float *v_means_add, gs_stream_t stream) {
    if (N == 0) return 0;
    GS_CHECK_ARG(means && viewmats && Ks && radii && conics && v_means2d && v_conics,
                 "null pointer");
    GS_CHECK_ARG(v_means2d_stride >= 2 && v_conics_stride >= 3, "bad gradient row strides");
    GS_CHECK_ARG((covars != nullptr) != (quats != nullptr && scales != nullptr),
                 "exactly one of covars / (quats, scales) must be given");
    GS_CHECK_ARG((v_compensations == nullptr) || (compensations != nullptr),
                 "v_compensations given without compensations");
    dim3 grid(gs_div_up(N, GS_BLOCK));
    const RowGrads rg = {nullptr, nullptr, nullptr, nullptr, nullptr, 0, 0, nullptr, nullptr, 0u, nullptr, nullptr, nullptr, 1.f, 0, nullptr};
    if (v_viewmats != nullptr) {
        hipLaunchKernelGGL(projection_bwd_kernel<true>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N,
                           means, covars, quats, scales, viewmats, Ks, image_width, image_height, eps2d,
                           camera_model, radii, conics, compensations, v_means2d, v_depths, v_conics,
                           v_compensations, v_means, v_covars, v_quats, v_scales, v_viewmats, v_means2d_stride,
                           v_conics_stride, v_means_add, rg);
    } else {
        hipLaunchKernelGGL(projection_bwd_kernel<false>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N,
                           means, covars, quats, scales, viewmats, Ks, image_width, image_height, eps2d,
                           camera_model, radii, conics, compensations, v_means2d, v_depths, v_conics,
                           v_compensations, v_means, v_covars, v_quats, v_scales, v_viewmats, v_means2d_stride,
                           v_conics_stride, v_means_add, rg);
    }
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_projection_packed_count(
    uint32_t C, uint32_t N, const float *means, const float *covars, const float *quats,
    const float *scales, const float *viewmats, const float *Ks, int32_t image_width,
    int32_t image_height, float eps2d, float near_plane, float far_plane, float radius_clip,
    int32_t camera_model, int32_t *block_cnts, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(means && viewmats && Ks && block_cnts, "null pointer");
    GS_CHECK_ARG((covars != nullptr) != (quats != nullptr && scales != nullptr),
                 "exactly one of covars / (quats, scales) must be given");
    dim3 grid(gs_div_up(N, GS_BLOCK), C);
    hipLaunchKernelGGL(projection_packed_kernel<false>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N,
                       means, covars, quats, scales, viewmats, Ks, image_width, image_height, eps2d,
                       near_plane, far_plane, radius_clip, camera_model, (const int32_t *)nullptr, block_cnts,
                       (int32_t *)nullptr, (int64_t *)nullptr, (int64_t *)nullptr, (int32_t *)nullptr,
                       (float *)nullptr, (float *)nullptr, (float *)nullptr, (float *)nullptr);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_projection_packed_fill(
    uint32_t C, uint32_t N, const float *means, const float *covars, const float *quats,
    const float *scales, const float *viewmats, const float *Ks, int32_t image_width,
    int32_t image_height, float eps2d, float near_plane, float far_plane, float radius_clip,
    int32_t camera_model, const int32_t *block_accum, int32_t *indptr, int64_t *camera_ids,
    int64_t *gaussian_ids, int32_t *radii, float *means2d, float *depths, float *conics,
    float *compensations, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(means && viewmats && Ks && block_accum && indptr, "null pointer");
    GS_CHECK_ARG((covars != nullptr) != (quats != nullptr && scales != nullptr),
                 "exactly one of covars / (quats, scales) must be given");
    dim3 grid(gs_div_up(N, GS_BLOCK), C);
    hipLaunchKernelGGL(projection_packed_kernel<true>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N,
                       means, covars, quats, scales, viewmats, Ks, image_width, image_height, eps2d,
                       near_plane, far_plane, radius_clip, camera_model, block_accum, (int32_t *)nullptr,
                       indptr, camera_ids, gaussian_ids, radii, means2d, depths, conics, compensations);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_projection_packed_bwd(
    uint32_t C, uint32_t N, uint32_t nnz, const float *means, const float *covars,
    const float *quats, const float *scales, const float *viewmats, const float *Ks,
    int32_t image_width, int32_t image_height, float eps2d, int32_t camera_model,
    const int64_t *camera_ids, const int64_t *gaussian_ids, const float *conics,
    const float *compensations, const float *v_means2d, const float *v_depths,
    const float *v_conics, const float *v_compensations, int32_t sparse_grad, float *v_means,
    float *v_covars, float *v_quats, float *v_scales, float *v_viewmats, gs_stream_t stream) {
    if (nnz == 0) return 0;
    GS_CHECK_ARG(means && viewmats && Ks, "null pointer");
    GS_CHECK_ARG((covars != nullptr) != (quats != nullptr && scales != nullptr),
                 "exactly one of covars / (quats, scales) must be given");
    GS_CHECK_ARG(camera_ids && gaussian_ids && conics && v_means2d && v_conics, "null pointer");
    dim3 grid(gs_div_up(nnz, GS_BLOCK));
    if (v_viewmats != nullptr) {
        hipLaunchKernelGGL(projection_packed_bwd_kernel<true>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream,
                           C, N, nnz, means, covars, quats, scales, viewmats, Ks, image_width, image_height,
                           eps2d, camera_model, camera_ids, gaussian_ids, conics, compensations, v_means2d,
                           v_depths, v_conics, v_compensations, sparse_grad, v_means, v_covars, v_quats,
                           v_scales, v_viewmats);
    } else {
        hipLaunchKernelGGL(projection_packed_bwd_kernel<false>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream,
                           C, N, nnz, means, covars, quats, scales, viewmats, Ks, image_width, image_height,
                           eps2d, camera_model, camera_ids, gaussian_ids, conics, compensations, v_means2d,
                           v_depths, v_conics, v_compensations, sparse_grad, v_means, v_covars, v_quats,
                           v_scales, v_viewmats);
    }
    GS_CHECK_LAUNCH();
    return 0;
}
