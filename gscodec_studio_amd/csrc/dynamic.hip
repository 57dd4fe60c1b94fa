// dynamic.hip -- temporal slicing of dynamic (spacetime) gaussians at one timestamp, fwd + bwd (gfx950).
//
// The step in front of the rasterizer for the reference's dynamic-scene trainer
// (examples/simple_trainer_dyngs.py:506-521, viewer examples/simple_viewer_dyn.py:84-101):
//     tau      = t - trbf_center                         (detached where it drives the motion)
//     trbf     = exp(-(tau / (sqrt(2) trbf_scale))^2)    temporal radial basis
//     opacity  = opacities * trbf
//     means_t  = means + m1 tau + m2 tau^2 + m3 tau^3    cubic motion, motion = [m1 | m2 | m3] (9 floats)
//     quats_t  = normalize(quats + tau omega)            F.normalize: x / max(|x|, 1e-12)
// ~25 elementwise torch kernels (each a pass over N x 3..9 floats) become one streaming kernel each
// way: forward reads 92 B and writes 36 B per splat, backward reads 128 B and writes 92 B.
// One lane per splat; rows are 12..36 B so accesses are partially coalesced, like the projection kernel.  The per-splat arithmetic lives
// in dynamic_dev.h (shared with projection_dyn.hip, which evaluates the slice inside the projection pass instead).
#include "gs_common.h"
#include "dynamic_dev.h"
#include "quant_dev.h"

namespace {

struct SliceArgs {
    uint32_t n;
    const float *means, *motion, *quats, *omega, *opacities, *trbf_center, *trbf_scale;
    float t;
};

__global__ void __launch_bounds__(GS_BLOCK) temporal_slice_fwd_kernel(SliceArgs a, float *__restrict__ means_t, float *__restrict__ quats_t,
                                                                      float *__restrict__ opacity_t, float *__restrict__ trbf_out) {
    GS_FP_STRICT;
    const uint32_t i = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const SliceTime st = slice_time(a.t, a.trbf_center[i], a.trbf_scale[i]);
    opacity_t[i] = (a.opacities[i] * st.trbf);
    if (trbf_out != nullptr) trbf_out[i] = st.trbf;
    const float tau = st.tau, t2 = (tau * tau), t3 = (t2 * tau);
    const float *m = a.motion + 9 * (size_t)i;
#pragma unroll
    for (int k = 0; k < 3; ++k) means_t[3 * (size_t)i + k] = slice_mean(a.means[3 * (size_t)i + k], m[k], m[3 + k], m[6 + k], tau, t2, t3);
    float qin[4], om[4], x[4], q[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        qin[k] = a.quats[4 * (size_t)i + k];
        om[k] = a.omega[4 * (size_t)i + k];
    }
    slice_quat(qin, om, tau, x, q);
#pragma unroll
    for (int k = 0; k < 4; ++k) quats_t[4 * (size_t)i + k] = q[k];
}

__global__ void __launch_bounds__(GS_BLOCK) temporal_slice_bwd_kernel(SliceArgs a, const float *__restrict__ v_means_t,
                                                                      const float *__restrict__ v_quats_t, const float *__restrict__ v_opacity_t,
                                                                      const float *__restrict__ v_trbf, float *__restrict__ v_means,
                                                                      float *__restrict__ v_motion, float *__restrict__ v_quats,
                                                                      float *__restrict__ v_omega, float *__restrict__ v_opacities,
                                                                      float *__restrict__ v_center, float *__restrict__ v_scale) {
    GS_FP_STRICT;
    const uint32_t i = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (i >= a.n) return;
    const float s = a.trbf_scale[i];
    const SliceTime st = slice_time(a.t, a.trbf_center[i], s);
    const float tau = st.tau;
    // opacity and the basis itself
    const float vo = v_opacity_t != nullptr ? v_opacity_t[i] : 0.f;
    float g_trbf = (vo * a.opacities[i]);
    if (v_trbf != nullptr) g_trbf = (g_trbf + v_trbf[i]);
    if (v_opacities != nullptr) v_opacities[i] = (vo * st.trbf);
    float vc, vs;
    slice_time_vjp(st, s, g_trbf, vc, vs);
    if (v_center != nullptr) v_center[i] = vc;
    if (v_scale != nullptr) v_scale[i] = vs;
    // motion: tau is detached there
    const float t2 = (tau * tau), t3 = (t2 * tau);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const float g = v_means_t != nullptr ? v_means_t[3 * (size_t)i + k] : 0.f;
        if (v_means != nullptr) v_means[3 * (size_t)i + k] = g;
        if (v_motion != nullptr) {
            v_motion[9 * (size_t)i + k] = (g * tau);
            v_motion[9 * (size_t)i + 3 + k] = (g * t2);
            v_motion[9 * (size_t)i + 6 + k] = (g * t3);
        }
    }
    float x[4], g[4], vx[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = (a.quats[4 * (size_t)i + k] + (tau * a.omega[4 * (size_t)i + k]));
        g[k] = v_quats_t != nullptr ? v_quats_t[4 * (size_t)i + k] : 0.f;
    }
    slice_quat_vjp(x, g, vx);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (v_quats != nullptr) v_quats[4 * (size_t)i + k] = vx[k];
        if (v_omega != nullptr) v_omega[4 * (size_t)i + k] = (vx[k] * tau);
    }
}

// The spacetime trainer's nine colour channels (examples/simple_trainer_STG.py:506-551):
//     colors_precomp = torch.cat((feature_color, feature_dir, tforpoly * feature_time), dim=1),  tforpoly = (t - trbf_center).detach()
// and, with the compression simulation on, the round-to-grid STE hook of each of the three parts in front (simulation.py:508-780,
// ops.py:57-75: the parameter is clamped IN PLACE, the output is the grid value, the gradient passes unchanged).
struct StgQuant {
    uint32_t mask; // bit p: part p (0 colours, 1 direction features, 2 time features) goes through the round quantizer
    float lo[3], hi[3], rng[3], qn[3];
};

// Four consecutive output elements per thread: one 16-byte store forward, one 16-byte load backward (the [N,9] array is 16-byte
// aligned and 36 N bytes long); the three [N,3] sources are touched in runs of three consecutive floats.
GS_DEV float stg_value(uint32_t e, float *__restrict__ colors, float *__restrict__ fdir, float *__restrict__ ftime,
                       const float *__restrict__ center, float t, const StgQuant &q) {
    GS_FP_STRICT;
    const uint32_t n = e / 9u, k = e - 9u * n, part = k / 3u;
    float *src = (part == 0u ? colors : part == 1u ? fdir : ftime) + 3u * (size_t)n + (k - 3u * part);
    float v = *src;
    if ((q.mask >> part) & 1u) {
        const float c = q_clamp(v, q.lo[part], q.hi[part]);
        if (!(c == v)) *src = c; // (rare; NaN stays NaN)
        v = q_round(c, q.lo[part], q.rng[part], q.qn[part]);
    }
    if (part == 2u) v = ((t - center[n]) * v);
    return v;
}

__global__ void __launch_bounds__(GS_BLOCK) stg_features_fwd_kernel(uint32_t n9, float *__restrict__ colors, float *__restrict__ fdir,
                                                                    float *__restrict__ ftime, const float *__restrict__ center, float t,
                                                                    StgQuant q, float *__restrict__ out) {
    const uint32_t e0 = (blockIdx.x * GS_BLOCK + threadIdx.x) * 4u;
    if (e0 >= n9) return;
    if (e0 + 4u <= n9) {
        float4 r;
        r.x = stg_value(e0, colors, fdir, ftime, center, t, q);
        r.y = stg_value(e0 + 1u, colors, fdir, ftime, center, t, q);
        r.z = stg_value(e0 + 2u, colors, fdir, ftime, center, t, q);
        r.w = stg_value(e0 + 3u, colors, fdir, ftime, center, t, q);
        *reinterpret_cast<float4 *>(out + e0) = r;
    } else {
        for (uint32_t e = e0; e < n9; ++e) out[e] = stg_value(e, colors, fdir, ftime, center, t, q);
    }
}

GS_DEV void stg_grad(uint32_t e, float v, const float *__restrict__ center, float t, float *__restrict__ v_colors, float *__restrict__ v_fdir,
                     float *__restrict__ v_ftime) {
    GS_FP_STRICT;
    const uint32_t n = e / 9u, k = e - 9u * n, part = k / 3u;
    float *dst = part == 0u ? v_colors : part == 1u ? v_fdir : v_ftime;
    if (dst == nullptr) return;
    if (part == 2u) v = ((t - center[n]) * v);
    dst[3u * (size_t)n + (k - 3u * part)] = v;
}

__global__ void __launch_bounds__(GS_BLOCK) stg_features_bwd_kernel(uint32_t n9, const float *__restrict__ v_out, const float *__restrict__ center,
                                                                    float t, float *__restrict__ v_colors, float *__restrict__ v_fdir,
                                                                    float *__restrict__ v_ftime) {
    const uint32_t e0 = (blockIdx.x * GS_BLOCK + threadIdx.x) * 4u;
    if (e0 >= n9) return;
    if (e0 + 4u <= n9) {
        const float4 v = *reinterpret_cast<const float4 *>(v_out + e0);
        stg_grad(e0, v.x, center, t, v_colors, v_fdir, v_ftime);
        stg_grad(e0 + 1u, v.y, center, t, v_colors, v_fdir, v_ftime);
        stg_grad(e0 + 2u, v.z, center, t, v_colors, v_fdir, v_ftime);
        stg_grad(e0 + 3u, v.w, center, t, v_colors, v_fdir, v_ftime);
    } else {
        for (uint32_t e = e0; e < n9; ++e) stg_grad(e, v_out[e], center, t, v_colors, v_fdir, v_ftime);
    }
}

// Tiled forms (every pointer 16-byte aligned -- whole tensors always are): a workgroup owns 256 gaussians; their three [256,3] source
// chunks and the [256,9] output chunk are contiguous, so every global access is a full 16-byte piece of a contiguous run and the 3 <-> 9
// interleave happens in LDS.  (The element-per-lane kernels above move the same bytes in 4-byte pieces: 59 / 40 us at 2 M gaussians
// against 40 / 27 here.)
__global__ void __launch_bounds__(GS_BLOCK) stg_features_fwd_tiled_kernel(uint32_t n, float *__restrict__ colors, float *__restrict__ fdir,
                                                                          float *__restrict__ ftime, const float *__restrict__ center, float t,
                                                                          StgQuant q, float *__restrict__ out) {
    GS_FP_STRICT;
    __shared__ __attribute__((aligned(16))) float s_out[GS_BLOCK * 9];
    __shared__ float s_tau[GS_BLOCK];
    const uint32_t n0 = blockIdx.x * GS_BLOCK, tid = threadIdx.x;
    const uint32_t cnt = min((uint32_t)GS_BLOCK, n - n0); // gaussians of this workgroup
    if (tid < cnt) s_tau[tid] = (t - center[n0 + tid]);
    __syncthreads();
    if (tid < (GS_BLOCK * 3) / 4) {
#pragma unroll
        for (uint32_t part = 0; part < 3u; ++part) {
            float *src = (part == 0u ? colors : part == 1u ? fdir : ftime) + 3u * (size_t)n0;
            const uint32_t f0 = 4u * tid;
            if (f0 >= 3u * cnt) continue;
            const bool full = f0 + 4u <= 3u * cnt;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (full) {
                const float4 r = *reinterpret_cast<const float4 *>(src + f0);
                v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
            } else {
                for (uint32_t j = 0; f0 + j < 3u * cnt; ++j) v[j] = src[f0 + j];
            }
            const bool quant = (q.mask >> part) & 1u;
            float c[4];
            bool changed = false;
#pragma unroll
            for (uint32_t j = 0; j < 4u; ++j) {
                c[j] = v[j];
                if (quant) {
                    c[j] = q_clamp(v[j], q.lo[part], q.hi[part]);
                    changed |= !(c[j] == v[j]);
                    v[j] = q_round(c[j], q.lo[part], q.rng[part], q.qn[part]);
                }
                const uint32_t f = f0 + j, nl = f / 3u, k = f - 3u * nl;
                if (f < 3u * cnt) s_out[nl * 9u + 3u * part + k] = part == 2u ? (s_tau[nl] * v[j]) : v[j];
            }
            if (changed) { // (rare: the parameter itself is clamped, as the hook does; NaN compares unequal and is stored back as itself)
                if (full) *reinterpret_cast<float4 *>(src + f0) = make_float4(c[0], c[1], c[2], c[3]);
                else for (uint32_t j = 0; f0 + j < 3u * cnt; ++j) src[f0 + j] = c[j];
            }
        }
    }
    __syncthreads();
    float *dst = out + 9u * (size_t)n0;
    for (uint32_t i = tid; 4u * i < 9u * cnt; i += GS_BLOCK) {
        if (4u * i + 4u <= 9u * cnt) reinterpret_cast<float4 *>(dst)[i] = reinterpret_cast<const float4 *>(s_out)[i];
        else for (uint32_t e = 4u * i; e < 9u * cnt; ++e) dst[e] = s_out[e];
    }
}

__global__ void __launch_bounds__(GS_BLOCK) stg_features_bwd_tiled_kernel(uint32_t n, const float *__restrict__ v_out, const float *__restrict__ center,
                                                                          float t, float *__restrict__ v_colors, float *__restrict__ v_fdir,
                                                                          float *__restrict__ v_ftime) {
    GS_FP_STRICT;
    __shared__ __attribute__((aligned(16))) float s_v[GS_BLOCK * 9];
    __shared__ float s_tau[GS_BLOCK];
    const uint32_t n0 = blockIdx.x * GS_BLOCK, tid = threadIdx.x;
    const uint32_t cnt = min((uint32_t)GS_BLOCK, n - n0);
    if (tid < cnt) s_tau[tid] = (t - center[n0 + tid]);
    const float *src = v_out + 9u * (size_t)n0;
    for (uint32_t i = tid; 4u * i < 9u * cnt; i += GS_BLOCK) {
        if (4u * i + 4u <= 9u * cnt) reinterpret_cast<float4 *>(s_v)[i] = reinterpret_cast<const float4 *>(src)[i];
        else for (uint32_t e = 4u * i; e < 9u * cnt; ++e) s_v[e] = src[e];
    }
    __syncthreads();
    if (tid >= (GS_BLOCK * 3) / 4) return;
#pragma unroll
    for (uint32_t part = 0; part < 3u; ++part) {
        float *dstp = part == 0u ? v_colors : part == 1u ? v_fdir : v_ftime;
        if (dstp == nullptr) continue;
        float *dst = dstp + 3u * (size_t)n0;
        const uint32_t f0 = 4u * tid;
        if (f0 >= 3u * cnt) continue;
        float v[4];
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            const uint32_t f = min(f0 + j, 3u * cnt - 1u), nl = f / 3u, k = f - 3u * nl;
            const float g = s_v[nl * 9u + 3u * part + k];
            v[j] = part == 2u ? (s_tau[nl] * g) : g;
        }
        if (f0 + 4u <= 3u * cnt) *reinterpret_cast<float4 *>(dst + f0) = make_float4(v[0], v[1], v[2], v[3]);
        else for (uint32_t j = 0; f0 + j < 3u * cnt; ++j) dst[f0 + j] = v[j];
    }
}

static inline bool stg_aligned(const void *a, const void *b, const void *c, const void *d) {
    return (((uintptr_t)a | (uintptr_t)b | (uintptr_t)c | (uintptr_t)d) % 16) == 0; // (a NULL pointer counts as aligned)
}

} // namespace

extern "C" int32_t gs_stg_features_fwd(uint32_t n, float *colors, float *features_dir, float *features_time, const float *trbf_center,
                                       float timestamp, uint32_t quant_mask, const float *quant_lo, const float *quant_hi,
                                       const float *quant_range, const float *quant_step_norm, float *out, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(colors && features_dir && features_time && trbf_center && out, "null pointer");
    GS_CHECK_ARG((uintptr_t)out % 16 == 0, "out must be 16-byte aligned");
    GS_CHECK_ARG((uint64_t)n * 9u < (1ull << 32), "n too large");
    GS_CHECK_ARG((quant_mask & 7u) == 0u || (quant_lo && quant_hi && quant_range && quant_step_norm),
                 "quant_mask set without the quantizer tables (3 floats each: colors, features_dir, features_time)");
    StgQuant q;
    q.mask = quant_mask & 7u;
    for (int p = 0; p < 3; ++p) {
        const bool on = (q.mask >> p) & 1u;
        q.lo[p] = on ? quant_lo[p] : 0.f; q.hi[p] = on ? quant_hi[p] : 0.f;
        q.rng[p] = on ? quant_range[p] : 1.f; q.qn[p] = on ? quant_step_norm[p] : 1.f;
    }
    if (stg_aligned(colors, features_dir, features_time, out))
        hipLaunchKernelGGL(stg_features_fwd_tiled_kernel, dim3(gs_div_up(n, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream, n, colors,
                           features_dir, features_time, trbf_center, timestamp, q, out);
    else
        hipLaunchKernelGGL(stg_features_fwd_kernel, dim3(gs_div_up(gs_div_up(9u * n, 4u), GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream, 9u * n, colors,
                           features_dir, features_time, trbf_center, timestamp, q, out);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_stg_features_bwd(uint32_t n, const float *v_out, const float *trbf_center, float timestamp, float *v_colors,
                                       float *v_features_dir, float *v_features_time, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(v_out && trbf_center, "null pointer");
    GS_CHECK_ARG((uintptr_t)v_out % 16 == 0, "v_out must be 16-byte aligned");
    GS_CHECK_ARG((uint64_t)n * 9u < (1ull << 32), "n too large");
    if (stg_aligned(v_colors, v_features_dir, v_features_time, v_out))
        hipLaunchKernelGGL(stg_features_bwd_tiled_kernel, dim3(gs_div_up(n, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream, n, v_out,
                           trbf_center, timestamp, v_colors, v_features_dir, v_features_time);
    else
        hipLaunchKernelGGL(stg_features_bwd_kernel, dim3(gs_div_up(gs_div_up(9u * n, 4u), GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream, 9u * n, v_out,
                           trbf_center, timestamp, v_colors, v_features_dir, v_features_time);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_temporal_slice_fwd(uint32_t n, const float *means, const float *motion, const float *quats, const float *omega,
                                         const float *opacities, const float *trbf_center, const float *trbf_scale, float timestamp,
                                         float *means_t, float *quats_t, float *opacity_t, float *trbf, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(means && motion && quats && omega && opacities && trbf_center && trbf_scale && means_t && quats_t && opacity_t,
                 "null pointer");
    SliceArgs a = {n, means, motion, quats, omega, opacities, trbf_center, trbf_scale, timestamp};
    hipLaunchKernelGGL(temporal_slice_fwd_kernel, dim3(gs_div_up(n, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream, a, means_t,
                       quats_t, opacity_t, trbf);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_temporal_slice_bwd(uint32_t n, const float *means, const float *motion, const float *quats, const float *omega,
                                         const float *opacities, const float *trbf_center, const float *trbf_scale, float timestamp,
                                         const float *v_means_t, const float *v_quats_t, const float *v_opacity_t, const float *v_trbf,
                                         float *v_means, float *v_motion, float *v_quats, float *v_omega, float *v_opacities,
                                         float *v_trbf_center, float *v_trbf_scale, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(quats && omega && opacities && trbf_center && trbf_scale, "null pointer");
    SliceArgs a = {n, means, motion, quats, omega, opacities, trbf_center, trbf_scale, timestamp};
    hipLaunchKernelGGL(temporal_slice_bwd_kernel, dim3(gs_div_up(n, GS_BLOCK)), dim3(GS_BLOCK), 0, (hipStream_t)stream, a, v_means_t,
                       v_quats_t, v_opacity_t, v_trbf, v_means, v_motion, v_quats, v_omega, v_opacities, v_trbf_center, v_trbf_scale);
    GS_CHECK_LAUNCH();
    return 0;
}
