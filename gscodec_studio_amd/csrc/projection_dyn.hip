// projection_dyn.hip -- the projection of DYNAMIC (spacetime) gaussians: the temporal slice at one timestamp -- and, opt-in, the
// trainer's activations and the round-to-grid quantizer hooks in front of it -- evaluated in the projection kernels' load phase (gfx950).
//
// The dynamic-scene trainer of the reference runs, per rendered frame (examples/simple_trainer_dyngs.py:463-554, BASELINE config 5):
//     hooks      STGCompressionSimulation: STE round quantizer on scales / quats / opacities / colors (...), `p + 0.` on the rest
//                (gsplat/compression_simulation/simulation.py:508-780, ops.py:57-75)
//     activate   scales = exp(.), opacities = sigmoid(.), trbf_scale = exp(.)                           (dyngs.py:493-505)
//     slice      tau = t - trbf_center; trbf = exp(-(tau / (sqrt2 trbf_scale))^2); opacity = opacities trbf;
//                means_t = means + m1 tau + m2 tau^2 + m3 tau^3; quats_t = normalize(quats + tau omega)   (dyngs.py:506-521)
//     render     rasterization(means_t, quats_t, scales, opacity, colors, ...)                          (dyngs.py:539-554)
// i.e. ~20 elementwise passes over N x (1..9) floats each way in front of a projection kernel that reads the result once.  At 2 M
// splats those passes are 230 us forward + 140 us backward of a 1.25 ms step (profiles/r06_dynamic_*).  Here the projection forward
// reads the RAW parameter rows (35 floats minus what it does not need), evaluates quantizer -> activation -> slice in registers (the
// arithmetic of quant_dev.h / dynamic_dev.h: bit-identical to the stand-alone kernels) and projects; the backward recomputes the same
// chain for the gaussians some camera saw and writes the gradients of the RAW parameters (motion, omega, trbf_* included) -- rows of
// unseen gaussians are never touched (prefilled zeros, as in gs_projection_rows_bwd).
//
// Layout / launch shape are those of projection.hip's row form: one lane per (camera, gaussian) forward, one lane per gaussian looping
// over cameras backward (no atomics, deterministic); splat rows out (include/gsplat_hip.h "Splat rows"), radii / depths densely,
// the tile count + per-workgroup sums for the binning in the same pass.
#include "gs_common.h"
#include "proj_models.h"
#include "projection_dev.h"
#include "quant_dev.h"
#include "dynamic_dev.h"

namespace {

// attribute slots of the in-kernel quantizer
enum { QA_SCALES = 0, QA_QUATS = 1, QA_OPACITIES = 2, QA_COLORS = 3 };

struct DynArgs {
    const float *motion, *omega, *center, *tscale; // [N,9] [N,4] [N] [N]
    float t;
    uint32_t raw;   // GS_DYN_RAW_* bits: scales are log-scales / opacities logits / trbf_scale a log-scale -- exp / sigmoid / exp in the kernel
    uint32_t quant; // bit k: attribute k goes through the STE round quantizer (clamped IN PLACE, ops.py:63) first
    float q_lo[4], q_hi[4], q_rng[4], q_n[4];
    float min_trbf;      // forward: gaussians whose temporal basis is <= this are culled at this timestamp (-1: none)
    uint8_t *alive_out;  // forward: [N] or NULL, 1 where trbf > min_trbf (the trainer's t_vis_mask)
};

// One gaussian's parameters after quantizer -> activation -> slice: what the projection consumes, and what the backward needs again.
struct DynSplat {
    float mx, my, mz;    // means_t
    float x[4], q[4];    // quats + tau omega (un-normalised) | normalised
    float s[3];          // activated scales
    float op_act, op_t;  // activated opacity | x trbf
    float ts;            // activated trbf_scale
    SliceTime st;
};

// The round quantizer on one value: c = the clamped parameter (what ops.py:63 leaves in the tensor), returns the grid value.
template <bool QUANT>
GS_DEV float dyn_q(const DynArgs &d, int a, float v, float &c) {
    GS_FP_STRICT;
    c = v;
    if (!QUANT || !((d.quant >> a) & 1u)) return v;
    c = q_clamp(v, d.q_lo[a], d.q_hi[a]);
    return q_round(c, d.q_lo[a], d.q_rng[a], d.q_n[a]);
}

// A row of K parameters through the quantizer: ALL loads first (one vector load), the clamped row stored back only when the clamp
// changed something (rare: the parameter itself is clamped in place, as the hooks do; NaN stays NaN)
template <bool QUANT, int K>
GS_DEV void dyn_q_row(const DynArgs &d, int a, float *__restrict__ p, bool write, float *out) {
    float v[K], c[K];
#pragma unroll
    for (int k = 0; k < K; ++k) v[k] = p[k];
    bool changed = false;
#pragma unroll
    for (int k = 0; k < K; ++k) {
        out[k] = dyn_q<QUANT>(d, a, v[k], c[k]);
        changed |= !(c[k] == v[k]);
    }
    if (QUANT && write && changed) {
#pragma unroll
        for (int k = 0; k < K; ++k) p[k] = c[k];
    }
}

GS_DEV void dyn_time(const DynArgs &d, uint32_t n, DynSplat &o) {
    float ts = d.tscale[n];
    if (d.raw & GS_DYN_RAW_TRBF_SCALE) ts = expf(ts);
    o.ts = ts;
    o.st = slice_time(d.t, d.center[n], ts);
}

GS_DEV void dyn_mean(const DynArgs &d, const float *__restrict__ means, uint32_t n, DynSplat &o) {
    GS_FP_STRICT;
    const float *p = means + 3 * (size_t)n;
    const float *m = d.motion + 9 * (size_t)n;
    const float tau = o.st.tau, t2 = (tau * tau), t3 = (t2 * tau);
    o.mx = slice_mean(p[0], m[0], m[3], m[6], tau, t2, t3);
    o.my = slice_mean(p[1], m[1], m[4], m[7], tau, t2, t3);
    o.mz = slice_mean(p[2], m[2], m[5], m[8], tau, t2, t3);
}

// quats / scales of gaussian n as the slice takes them: through the round quantizer when hooked (raw[0..3] quaternion, raw[4..6] scales)
template <bool QUANT>
GS_DEV void dyn_shape_load(const DynArgs &d, float *__restrict__ quats, float *__restrict__ scales, uint32_t n, bool write, float raw[7]) {
    dyn_q_row<QUANT, 4>(d, QA_QUATS, quats + 4 * (size_t)n, write, raw);
    dyn_q_row<QUANT, 3>(d, QA_SCALES, scales + 3 * (size_t)n, write, raw + 4);
}

// ... and their slice + activation (ONE body for every instance: the quantized and the plain kernels round the same way)
GS_DEV void dyn_shape_eval(const DynArgs &d, const float raw[7], uint32_t n, DynSplat &o) {
    float om[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) om[k] = d.omega[4 * (size_t)n + k];
    slice_quat(raw, om, o.st.tau, o.x, o.q);
#pragma unroll
    for (int k = 0; k < 3; ++k) o.s[k] = (d.raw & GS_DYN_RAW_SCALES) ? expf(raw[4 + k]) : raw[4 + k];
}

template <bool QUANT>
GS_DEV void dyn_opacity(const DynArgs &d, float *__restrict__ opacities, uint32_t n, bool write, DynSplat &o) {
    GS_FP_STRICT;
    float v;
    dyn_q_row<QUANT, 1>(d, QA_OPACITIES, opacities + n, write, &v);
    if (d.raw & GS_DYN_RAW_OPACITIES) v = q_act<GS_ACT_SIGMOID>(v);
    o.op_act = v;
    o.op_t = (v * o.st.trbf);
}

struct DynRowArgs {
    float *opacities; // [N] or NULL
    float *colors;    // [N,3] or NULL (more than three channels: the caller's own colour array)
    int antialiased;
    int32_t *tiles_per_gauss;
    int32_t *block_sums;
    float tile_size;
    int32_t tile_width, tile_height;
};

template <bool QUANT>
__global__ void __launch_bounds__(GS_BLOCK) projection_dyn_fwd_kernel(
    uint32_t C, uint32_t N, const float *__restrict__ means, float *__restrict__ quats, float *__restrict__ scales,
    const float *__restrict__ viewmats, const float *__restrict__ Ks, int W, int H, float eps2d, float near_plane, float far_plane,
    float radius_clip, int camera_model, int32_t *__restrict__ radii, float *__restrict__ rows, float *__restrict__ depths, DynRowArgs rx,
    DynArgs dyn) {
    const uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    const uint32_t c = blockIdx.y;
    if (rx.tiles_per_gauss == nullptr && n >= N) return;
    const bool in = n < N;
    const bool write = c == 0u; // the in-place clamp of the quantizer: once per gaussian
    Camera cam = load_camera(viewmats, Ks, c);
    Splat2D s;
    s.radius = 0;
    DynSplat o;
    float col[3] = {0.f, 0.f, 0.f};
    if (in) {
        dyn_time(dyn, n, o);
        // temporal visibility (simple_trainer_dyngs.py:526-535 filters these splats out before rasterization): culled here instead
        const bool alive = o.st.trbf > dyn.min_trbf;
        if (dyn.alive_out != nullptr && write) dyn.alive_out[n] = alive ? 1 : 0;
        dyn_mean(dyn, means, n, o);
        float raw[7];
        if (QUANT) {
            // the quantizer clamps the PARAMETERS of every gaussian, seen or not: all hooked rows are read (and the rare
            // out-of-range value stored back) before anything is culled
            dyn_shape_load<true>(dyn, quats, scales, n, write, raw);
            if (rx.opacities != nullptr) dyn_opacity<true>(dyn, rx.opacities, n, write, o);
            if (rx.colors != nullptr) dyn_q_row<true, 3>(dyn, QA_COLORS, rx.colors + 3 * (size_t)n, write, col);
        }
        s = project_point<false>(cam, o.mx, o.my, o.mz, [&]() {
            if (!QUANT) dyn_shape_load<false>(dyn, quats, scales, n, false, raw);
            dyn_shape_eval(dyn, raw, n, o);
            return covar_from_rot_scale(quat_to_rotmat(o.q[0], o.q[1], o.q[2], o.q[3]), o.s[0], o.s[1], o.s[2]); },
            W, H, eps2d, near_plane, far_plane, radius_clip, camera_model);
        // (culled AFTER the projection, not around it: the projection chain is contractible code shared with projection.hip, and it is
        // the unchanged shape of this call that keeps the two kernels' rows bit-identical -- tests/test_gpu_dynamic_fused.py watches it)
        if (!alive) s.radius = 0;
    }
    const size_t idx = (size_t)c * N + n;
    if (rx.tiles_per_gauss != nullptr) { // (uniform)
        rows_count_tiles(s, in, idx, rx.tiles_per_gauss, rx.block_sums, rx.tile_size, rx.tile_width, rx.tile_height);
        if (!in) return;
    }
    radii[idx] = s.radius;
    if (s.radius <= 0) return;
    float *row = rows + GS_ROW_FLOATS * idx;
    depths[idx] = s.depth;
    float op = 0.f;
    if (rx.opacities != nullptr) {
        if (!QUANT) dyn_opacity<false>(dyn, rx.opacities, n, false, o);
        op = o.op_t;
    }
    if (rx.antialiased) op *= s.comp;
    reinterpret_cast<float4 *>(row)[0] = make_float4(s.mx, s.my, s.ca, s.cb);
    if (rx.colors != nullptr) {
        if (!QUANT) {
            const float *cp = rx.colors + 3 * (size_t)n;
            col[0] = cp[0]; col[1] = cp[1]; col[2] = cp[2];
        }
        reinterpret_cast<float4 *>(row)[1] = make_float4(s.cc, op, col[0], col[1]);
        reinterpret_cast<float4 *>(row)[2] = make_float4(col[2], s.depth, __int_as_float(s.radius), s.comp);
    } else {
        reinterpret_cast<float2 *>(row)[2] = make_float2(s.cc, op);
        row[GS_ROW_DEPTH] = s.depth;
        reinterpret_cast<float2 *>(row)[5] = make_float2(__int_as_float(s.radius), s.comp);
    }
}

struct DynGradOut {
    float *v_means, *v_quats, *v_scales, *v_motion, *v_omega, *v_center, *v_tscale, *v_opacities, *v_colors;
    int prefilled; // every output holds zeros already: gaussians no camera sees are not stored
};

template <bool QUANT>
__global__ void __launch_bounds__(GS_BLOCK) projection_dyn_bwd_kernel(
    uint32_t C, uint32_t N, const float *__restrict__ means, const float *__restrict__ quats, const float *__restrict__ scales,
    const float *__restrict__ opacities, const float *__restrict__ viewmats, const float *__restrict__ Ks, int W, int H, float eps2d,
    int camera_model, const int32_t *__restrict__ radii, const float *__restrict__ rows, const float *__restrict__ grad_rows,
    const float *__restrict__ v_depths, int antialiased, DynArgs dyn, DynGradOut out) {
    const uint32_t n = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (n >= N) return;
    DynSplat o;
    Sym3 S = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    ProjGrad g;
    grad_zero(g);
    float v_op = 0.f, v_c0 = 0.f, v_c1 = 0.f, v_c2 = 0.f;
    bool any = false;
    for (uint32_t c = 0; c < C; ++c) {
        const size_t idx = (size_t)c * N + n;
        if (radii[idx] <= 0) continue;
        if (!any) {
            // quantizer -> activation -> slice, as the forward evaluated them (the parameters were clamped by the forward: the clamp
            // here is the identity, nothing is stored)
            dyn_time(dyn, n, o);
            dyn_mean(dyn, means, n, o);
            float raw[7];
            dyn_shape_load<QUANT>(dyn, const_cast<float *>(quats), const_cast<float *>(scales), n, false, raw);
            dyn_shape_eval(dyn, raw, n, o);
            o.op_act = o.op_t = 0.f;
            if (opacities != nullptr) dyn_opacity<QUANT>(dyn, const_cast<float *>(opacities), n, false, o);
            S = covar_from_rot_scale(quat_to_rotmat(o.q[0], o.q[1], o.q[2], o.q[3]), o.s[0], o.s[1], o.s[2]);
            any = true;
        }
        Camera cam = load_camera(viewmats, Ks, c);
        const float4 *r = reinterpret_cast<const float4 *>(rows + GS_ROW_FLOATS * idx);
        const float4 *gr = reinterpret_cast<const float4 *>(grad_rows + GS_ROW_FLOATS * idx);
        const float4 r0 = r[0], g0 = gr[0], g1 = gr[1];
        const float cc = reinterpret_cast<const float *>(r)[GS_ROW_CONIC + 2];
        const float comp = antialiased ? reinterpret_cast<const float *>(r)[GS_ROW_COMPENSATION] : 1.f;
        const float v_opac_cn = g1.y;
        const float v_comp = antialiased ? v_opac_cn * o.op_t : 0.f;
        v_op += v_opac_cn * comp;
        v_c0 += g1.z;
        v_c1 += g1.w;
        if (out.v_colors != nullptr) v_c2 += reinterpret_cast<const float *>(gr)[GS_ROW_COLOR + 2];
        project_one_vjp<false>(cam, o.mx, o.my, o.mz, S, W, H, eps2d, camera_model, r0.z, r0.w, cc, comp, v_comp, antialiased != 0, g0.x, g0.y,
                               v_depths != nullptr ? v_depths[idx] : 0.f, g0.z, g0.w, g1.x, g);
    }
    if (!any && out.prefilled) return;
    float vq[4] = {0.f, 0.f, 0.f, 0.f}, vs[3] = {0.f, 0.f, 0.f}, vx[4] = {0.f, 0.f, 0.f, 0.f};
    float tau = 0.f, t2 = 0.f, t3 = 0.f, v_opac = 0.f, v_center = 0.f, v_tscale = 0.f;
    if (any) {
        const Mat3 R = quat_to_rotmat(o.q[0], o.q[1], o.q[2], o.q[3]);
        covar_vjp_quat_scale(o.q[0], o.q[1], o.q[2], o.q[3], o.s[0], o.s[1], o.s[2], R, sym3_to_mat3(g.v_S), vq, vs);
        // ---- back through the slice (dynamic_dev.h; what gs_temporal_slice_bwd computes from v_means_t / v_quats_t / v_opacity_t)
        tau = o.st.tau; t2 = (tau * tau); t3 = (t2 * tau);
        slice_quat_vjp(o.x, vq, vx);
        v_opac = (v_op * o.st.trbf);
        slice_time_vjp(o.st, o.ts, (v_op * o.op_act), v_center, v_tscale);
        // ---- back through the activations (the round STE in between is the identity, ops.py:73-75)
        if (dyn.raw & GS_DYN_RAW_SCALES) {
#pragma unroll
            for (int k = 0; k < 3; ++k) vs[k] = q_act_grad<GS_ACT_EXP>(o.s[k], vs[k]);
        }
        if (dyn.raw & GS_DYN_RAW_OPACITIES) v_opac = q_act_grad<GS_ACT_SIGMOID>(o.op_act, v_opac);
        if (dyn.raw & GS_DYN_RAW_TRBF_SCALE) v_tscale = q_act_grad<GS_ACT_EXP>(o.ts, v_tscale);
    }
    const float gm[3] = {g.v_px, g.v_py, g.v_pz};
    if (out.v_means != nullptr) {
#pragma unroll
        for (int k = 0; k < 3; ++k) out.v_means[3 * (size_t)n + k] = gm[k];
    }
    if (out.v_motion != nullptr) {
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            out.v_motion[9 * (size_t)n + k] = (gm[k] * tau);
            out.v_motion[9 * (size_t)n + 3 + k] = (gm[k] * t2);
            out.v_motion[9 * (size_t)n + 6 + k] = (gm[k] * t3);
        }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (out.v_quats != nullptr) out.v_quats[4 * (size_t)n + k] = vx[k];
        if (out.v_omega != nullptr) out.v_omega[4 * (size_t)n + k] = (vx[k] * tau);
    }
    if (out.v_scales != nullptr) {
#pragma unroll
        for (int k = 0; k < 3; ++k) out.v_scales[3 * (size_t)n + k] = vs[k];
    }
    if (out.v_opacities != nullptr) out.v_opacities[n] = v_opac;
    if (out.v_center != nullptr) out.v_center[n] = v_center;
    if (out.v_tscale != nullptr) out.v_tscale[n] = v_tscale;
    if (out.v_colors != nullptr) {
        out.v_colors[3 * (size_t)n] = v_c0;
        out.v_colors[3 * (size_t)n + 1] = v_c1;
        out.v_colors[3 * (size_t)n + 2] = v_c2;
    }
}

int dyn_args(DynArgs &d, const float *motion, const float *omega, const float *trbf_center, const float *trbf_scale, float timestamp,
             uint32_t raw_params, uint32_t quant_mask, const float *quant_lo, const float *quant_hi, const float *quant_range,
             const float *quant_step_norm) {
    d.motion = motion; d.omega = omega; d.center = trbf_center; d.tscale = trbf_scale;
    d.t = timestamp; d.raw = (uint32_t)raw_params & 7u; d.quant = quant_mask & 15u;
    d.min_trbf = -1.f; d.alive_out = nullptr;
    for (int k = 0; k < 4; ++k) {
        const bool on = (d.quant >> k) & 1u;
        if (on && !(quant_lo && quant_hi && quant_range && quant_step_norm)) return 1;
        d.q_lo[k] = on ? quant_lo[k] : 0.f; d.q_hi[k] = on ? quant_hi[k] : 0.f;
        d.q_rng[k] = on ? quant_range[k] : 1.f; d.q_n[k] = on ? quant_step_norm[k] : 1.f;
    }
    return 0;
}

} // namespace

extern "C" int32_t gs_projection_rows_dyn_fwd(
    uint32_t C, uint32_t N, const float *means, float *quats, float *scales, const float *motion, const float *omega,
    const float *trbf_center, const float *trbf_scale, float timestamp, float min_trbf, uint8_t *trbf_alive, uint32_t raw_params,
    uint32_t quant_mask, const float *quant_lo, const float *quant_hi, const float *quant_range, const float *quant_step_norm,
    const float *viewmats, const float *Ks, int32_t image_width, int32_t image_height, float eps2d, float near_plane, float far_plane,
    float radius_clip, int32_t camera_model, float *opacities, float *colors, int32_t antialiased, uint32_t tile_size, uint32_t tile_width,
    uint32_t tile_height, int32_t *tiles_per_gauss, int32_t *block_sums, int32_t *radii, float *depths, float *rows, gs_stream_t stream) {
    if (C == 0 || N == 0) return 0;
    GS_CHECK_ARG(means && quats && scales && motion && omega && trbf_center && trbf_scale && viewmats && Ks && radii && depths && rows,
                 "null pointer");
    GS_CHECK_ARG((uintptr_t)rows % 64 == 0, "the row buffer must be 64-byte aligned");
    GS_CHECK_ARG(camera_model >= 0 && camera_model <= 2, "bad camera_model");
    GS_CHECK_ARG(!antialiased || opacities != nullptr, "antialiased needs the opacities");
    GS_CHECK_ARG(tiles_per_gauss != nullptr || block_sums == nullptr, "block_sums come with tiles_per_gauss");
    GS_CHECK_ARG(tiles_per_gauss == nullptr || tile_size > 0, "tile_size must be > 0");
    GS_CHECK_ARG(!((quant_mask >> 2) & 1u) || opacities != nullptr, "quantized opacities need the opacities");
    GS_CHECK_ARG(!((quant_mask >> 3) & 1u) || colors != nullptr, "quantized colors need the colors");
    DynArgs d;
    GS_CHECK_ARG(dyn_args(d, motion, omega, trbf_center, trbf_scale, timestamp, raw_params, quant_mask, quant_lo, quant_hi, quant_range,
                          quant_step_norm) == 0, "quant_mask set without the quantizer tables (4 floats each: scales, quats, opacities, colors)");
    d.min_trbf = min_trbf;
    d.alive_out = trbf_alive;
    const DynRowArgs rx = {opacities, colors, antialiased, tiles_per_gauss, block_sums, (float)tile_size, (int32_t)tile_width, (int32_t)tile_height};
    const dim3 grid(gs_div_up(N, GS_BLOCK), C);
    if (d.quant)
        hipLaunchKernelGGL(projection_dyn_fwd_kernel<true>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N, means, quats, scales, viewmats, Ks,
                           image_width, image_height, eps2d, near_plane, far_plane, radius_clip, camera_model, radii, rows, depths, rx, d);
    else
        hipLaunchKernelGGL(projection_dyn_fwd_kernel<false>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N, means, quats, scales, viewmats, Ks,
                           image_width, image_height, eps2d, near_plane, far_plane, radius_clip, camera_model, radii, rows, depths, rx, d);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_projection_rows_dyn_bwd(
    uint32_t C, uint32_t N, const float *means, const float *quats, const float *scales, const float *motion, const float *omega,
    const float *trbf_center, const float *trbf_scale, float timestamp, uint32_t raw_params, uint32_t quant_mask, const float *quant_lo,
    const float *quant_hi, const float *quant_range, const float *quant_step_norm, const float *viewmats, const float *Ks,
    int32_t image_width, int32_t image_height, float eps2d, int32_t camera_model, const int32_t *radii, const float *rows,
    const float *grad_rows, const float *v_depths, const float *opacities, int32_t antialiased, float *v_means, float *v_quats,
    float *v_scales, float *v_motion, float *v_omega, float *v_trbf_center, float *v_trbf_scale, float *v_opacities, float *v_colors,
    int32_t outputs_prefilled, gs_stream_t stream) {
    if (N == 0) return 0;
    GS_CHECK_ARG(means && quats && scales && motion && omega && trbf_center && trbf_scale && viewmats && Ks && radii && rows && grad_rows,
                 "null pointer");
    GS_CHECK_ARG((uintptr_t)rows % 16 == 0 && (uintptr_t)grad_rows % 16 == 0, "row buffers must be 16-byte aligned");
    GS_CHECK_ARG(camera_model >= 0 && camera_model <= 2, "bad camera_model");
    GS_CHECK_ARG(opacities != nullptr || (!antialiased && v_opacities == nullptr && v_trbf_center == nullptr && v_trbf_scale == nullptr),
                 "the opacity / trbf gradients (and antialiased) need the opacities");
    DynArgs d;
    GS_CHECK_ARG(dyn_args(d, motion, omega, trbf_center, trbf_scale, timestamp, raw_params, quant_mask, quant_lo, quant_hi, quant_range,
                          quant_step_norm) == 0, "quant_mask set without the quantizer tables");
    d.quant &= opacities != nullptr ? 15u : 11u;
    const DynGradOut out = {v_means, v_quats, v_scales, v_motion, v_omega, v_trbf_center, v_trbf_scale, v_opacities, v_colors, outputs_prefilled != 0};
    const dim3 grid(gs_div_up(N, GS_BLOCK));
    if (d.quant)
        hipLaunchKernelGGL(projection_dyn_bwd_kernel<true>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N, means, quats, scales, opacities,
                           viewmats, Ks, image_width, image_height, eps2d, camera_model, radii, rows, grad_rows, v_depths, antialiased, d, out);
    else
        hipLaunchKernelGGL(projection_dyn_bwd_kernel<false>, grid, dim3(GS_BLOCK), 0, (hipStream_t)stream, C, N, means, quats, scales, opacities,
                           viewmats, Ks, image_width, image_height, eps2d, camera_model, radii, rows, grad_rows, v_depths, antialiased, d, out);
    GS_CHECK_LAUNCH();
    return 0;
}
