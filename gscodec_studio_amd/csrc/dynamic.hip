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

} // namespace

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
