// dynamic_dev.h -- the temporal slice of one dynamic (spacetime) gaussian at one timestamp and its VJP, shared by dynamic.hip (the
// stand-alone slice kernels) and projection_dyn.hip (the slice evaluated in the projection's load phase): ONE definition, no fma
// contraction (GS_FP_STRICT), so that slice-then-project and the fused kernels see bit-identical means / quaternions / opacities.
//   reference: examples/simple_trainer_dyngs.py:506-521 (viewer: examples/simple_viewer_dyn.py:84-101)
//     tau      = t - trbf_center                         (detached where it drives the motion: `tforpoly`)
//     trbf     = exp(-(tau / (sqrt(2) trbf_scale))^2)    temporal radial basis
//     opacity  = opacities * trbf
//     means_t  = means + m1 tau + m2 tau^2 + m3 tau^3    cubic motion, motion = [m1 | m2 | m3] (9 floats)
//     quats_t  = normalize(quats + tau omega)            F.normalize: x / max(|x|, 1e-12)
// Device code only.
#pragma once
#include "gs_common.h"

namespace {

#define GS_SQRT2F 1.4142135623730951f

struct SliceTime {
    float tau, d, trbf; // t - centre, tau / (sqrt2 scale), exp(-d^2)
};

GS_DEV SliceTime slice_time(float t, float center, float scale) {
    GS_FP_STRICT;
    SliceTime s;
    s.tau = (t - center);
    s.d = (s.tau / (GS_SQRT2F * scale));
    s.trbf = expf(-(s.d * s.d));
    return s;
}

// mean + m1 tau + m2 tau^2 + m3 tau^3 for one coordinate (the reference's left-to-right sum; tau^3 as (tau tau) tau)
GS_DEV float slice_mean(float mean, float m1, float m2, float m3, float tau, float t2, float t3) {
    GS_FP_STRICT;
    return (((mean + (m1 * tau)) + (m2 * t2)) + (m3 * t3));
}

// q = normalize(quat + tau omega); x[] keeps the un-normalised sum (the backward needs it), returns 1 / max(|x|, 1e-12)
GS_DEV float slice_quat(const float quat[4], const float omega[4], float tau, float x[4], float q[4]) {
    GS_FP_STRICT;
    float nn = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        x[k] = (quat[k] + (tau * omega[k]));
        nn = (nn + (x[k] * x[k]));
    }
    const float inv = (1.f / fmaxf(sqrtf(nn), 1e-12f));
#pragma unroll
    for (int k = 0; k < 4; ++k) q[k] = (x[k] * inv);
    return inv;
}

// VJP of the time part: g_trbf = d L / d trbf (opacity path: v_opacity_t * opacities, plus a direct v_trbf).
//   d trbf / d d = -2 d trbf;  d d / d centre = -1 / (sqrt2 s);  d d / d s = -d / s
GS_DEV void slice_time_vjp(const SliceTime &s, float scale, float g_trbf, float &v_center, float &v_scale) {
    GS_FP_STRICT;
    const float g_d = (g_trbf * ((-2.f * s.d) * s.trbf));
    v_center = (-g_d / (GS_SQRT2F * scale));
    v_scale = ((-g_d * s.d) / scale);
}

// VJP of y = x / max(|x|, eps):  v_x = (g - y (y . g)) / |x|   (eps branch: g / eps)
GS_DEV void slice_quat_vjp(const float x[4], const float g[4], float vx[4]) {
    GS_FP_STRICT;
    float nn = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) nn = (nn + (x[k] * x[k]));
    const float len = sqrtf(nn);
    if (len > 1e-12f) {
        const float inv = (1.f / len);
        float dot = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) dot = (dot + ((x[k] * inv) * g[k]));
#pragma unroll
        for (int k = 0; k < 4; ++k) vx[k] = ((g[k] - ((x[k] * inv) * dot)) * inv);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) vx[k] = (g[k] * 1e12f);
    }
}

} // namespace
