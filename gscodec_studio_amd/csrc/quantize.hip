// quantize.hip -- Q1/Q2: per-splat quantize/dequantize straight-through estimators, and the min-max
// grid quantizer of the on-disk attribute format (gfx950).
//
// Replaces the elementwise torch chains of
//   gsplat/compression_simulation/ops.py:39-54 (fake_quantize_ste, "noise" mode)
//   gsplat/compression_simulation/ops.py:57-75 (STE, "round" mode)
// with one streaming pass each (16-byte loads/stores; pure HBM bandwidth).
// Arithmetic is IEEE fp32, no contraction, in the reference's operation order, so the
// outputs are bit-identical to the torch ops (file compiled with -ffp-contract=off).
#include "gs_common.h"
#include "quant_dev.h"

namespace {

constexpr int Q_VEC = 4;

GS_DEV float q_noise(float x, float nz, float lo, float hi, float q_step) {
    return __fadd_rn(q_clamp(x, lo, hi), __fmul_rn(nz, q_step));
}

template <int ACT>
__global__ void __launch_bounds__(GS_BLOCK) quant_noise_fwd_kernel(
    uint64_t n, const float *__restrict__ x, const float *__restrict__ noise, float lo, float hi,
    float q_step, float *__restrict__ out, int vec_ok) {
    uint64_t stride = (uint64_t)gridDim.x * GS_BLOCK;
    uint64_t t = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    uint64_t nv = vec_ok ? n / Q_VEC : 0;
    for (uint64_t i = t; i < nv; i += stride) {
        float4 a = reinterpret_cast<const float4 *>(x)[i];
        float4 z = reinterpret_cast<const float4 *>(noise)[i];
        float4 r;
        r.x = q_act<ACT>(q_noise(a.x, z.x, lo, hi, q_step));
        r.y = q_act<ACT>(q_noise(a.y, z.y, lo, hi, q_step));
        r.z = q_act<ACT>(q_noise(a.z, z.z, lo, hi, q_step));
        r.w = q_act<ACT>(q_noise(a.w, z.w, lo, hi, q_step));
        reinterpret_cast<float4 *>(out)[i] = r;
    }
    for (uint64_t i = nv * Q_VEC + t; i < n; i += stride) out[i] = q_act<ACT>(q_noise(x[i], noise[i], lo, hi, q_step));
}

GS_DEV float q_mask(float x, float v, float lo, float hi) {
    // autograd of torch.clamp: gradient passes where lo <= x <= hi
    return (x >= lo && x <= hi) ? v : 0.f;
}

// MASK: the clamp's gradient gate of the noise mode (the round mode passes gradients everywhere, ops.py:73-75)
template <int ACT, bool MASK>
__global__ void __launch_bounds__(GS_BLOCK) quant_noise_bwd_kernel(
    uint64_t n, const float *__restrict__ x, const float *__restrict__ v_out, const float *__restrict__ out, float lo, float hi,
    float *__restrict__ v_x, int vec_ok) {
    uint64_t stride = (uint64_t)gridDim.x * GS_BLOCK;
    uint64_t t = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    uint64_t nv = vec_ok ? n / Q_VEC : 0;
    for (uint64_t i = t; i < nv; i += stride) {
        float4 a = MASK ? reinterpret_cast<const float4 *>(x)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
        float4 g = reinterpret_cast<const float4 *>(v_out)[i];
        if (ACT != 0) {
            const float4 o = reinterpret_cast<const float4 *>(out)[i];
            g.x = q_act_grad<ACT>(o.x, g.x); g.y = q_act_grad<ACT>(o.y, g.y);
            g.z = q_act_grad<ACT>(o.z, g.z); g.w = q_act_grad<ACT>(o.w, g.w);
        }
        float4 r = g;
        if (MASK) {
            r.x = q_mask(a.x, g.x, lo, hi);
            r.y = q_mask(a.y, g.y, lo, hi);
            r.z = q_mask(a.z, g.z, lo, hi);
            r.w = q_mask(a.w, g.w, lo, hi);
        }
        reinterpret_cast<float4 *>(v_x)[i] = r;
    }
    for (uint64_t i = nv * Q_VEC + t; i < n; i += stride) {
        float g = v_out[i];
        if (ACT != 0) g = q_act_grad<ACT>(out[i], g);
        v_x[i] = MASK ? q_mask(x[i], g, lo, hi) : g;
    }
}

template <int ACT>
__global__ void __launch_bounds__(GS_BLOCK) quant_round_fwd_kernel(
    uint64_t n, float *__restrict__ x, float lo, float hi, float range, float qn,
    float *__restrict__ out, int vec_ok) {
    uint64_t stride = (uint64_t)gridDim.x * GS_BLOCK;
    uint64_t t = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x;
    uint64_t nv = vec_ok ? n / Q_VEC : 0;
    for (uint64_t i = t; i < nv; i += stride) {
        const float4 a0 = reinterpret_cast<const float4 *>(x)[i];
        float4 a;
        a.x = q_clamp(a0.x, lo, hi); a.y = q_clamp(a0.y, lo, hi);
        a.z = q_clamp(a0.z, lo, hi); a.w = q_clamp(a0.w, lo, hi);
        // in-place clamp of the parameter (ops.py:63) -- stored only where it changes a value (a third of the kernel's traffic otherwise)
        if (a.x != a0.x || a.y != a0.y || a.z != a0.z || a.w != a0.w) reinterpret_cast<float4 *>(x)[i] = a;
        float4 r;
        r.x = q_act<ACT>(q_round(a.x, lo, range, qn)); r.y = q_act<ACT>(q_round(a.y, lo, range, qn));
        r.z = q_act<ACT>(q_round(a.z, lo, range, qn)); r.w = q_act<ACT>(q_round(a.w, lo, range, qn));
        reinterpret_cast<float4 *>(out)[i] = r;
    }
    for (uint64_t i = nv * Q_VEC + t; i < n; i += stride) {
        const float a0 = x[i], a = q_clamp(a0, lo, hi);
        if (a != a0) x[i] = a;
        out[i] = q_act<ACT>(q_round(a, lo, range, qn));
    }
}

uint32_t stream_grid(uint64_t n) {
    uint64_t blocks = (n / Q_VEC + GS_BLOCK - 1) / GS_BLOCK;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 8) blocks = 256 * 8; // 8 workgroups per CU, grid-stride the rest
    return (uint32_t)blocks;
}

bool aligned16(const void *a, const void *b, const void *c) {
    return ((uintptr_t)a % 16 == 0) && ((uintptr_t)b % 16 == 0) && ((uintptr_t)c % 16 == 0);
}


// ---------------------------------------------------------------------------
// On-disk attribute format: per-channel min-max quantization of a [rows, channels] grid
// to 8 / k (<= 8) / 16 bits, and its exact inverse.  Reference arithmetic
// (gsplat/compression/png_compression.py:186-193, 260-268, 332-345 encode; 224-232, 298-306, 375-389 decode):
//   encode: norm = (x - min) / (max - min) in fp32 (torch), img = round(norm * (2^q - 1)) in fp32 with
//           round-half-to-even (numpy), k-bit images are shifted left by 8 - q, 16-bit ones split in two planes;
//   decode: norm = img / (2^q - 1) in float64 (numpy), grid = norm * (max - min) + min with (max - min) formed in
//           fp32 and the product / sum in float64 (torch type promotion), then cast to fp32.
// NaN (max == min) encodes as 0, like numpy's float -> uint8 cast on x86.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(GS_BLOCK) grid_quantize_kernel(uint64_t n, uint32_t channels, const float *__restrict__ x,
                                                                 const float *__restrict__ mins, const float *__restrict__ maxs,
                                                                 float levels, uint32_t shift, uint8_t *__restrict__ lo,
                                                                 uint8_t *__restrict__ hi) {
    const uint64_t stride = (uint64_t)gridDim.x * GS_BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x; i < n; i += stride) {
        const uint32_t c = (uint32_t)(i % channels);
        const float mn = mins[c];
        const float norm = __fdiv_rn(__fsub_rn(x[i], mn), __fsub_rn(maxs[c], mn));
        const float r = rintf(__fmul_rn(norm, levels));
        uint32_t q = (r == r) ? (uint32_t)fminf(fmaxf(r, 0.f), 65535.f) : 0u;
        if (hi != nullptr) {
            lo[i] = (uint8_t)(q & 0xFFu);
            hi[i] = (uint8_t)((q >> 8) & 0xFFu);
        } else {
            lo[i] = (uint8_t)((q & 0xFFu) << shift);
        }
    }
}

__global__ void __launch_bounds__(GS_BLOCK) grid_dequantize_kernel(uint64_t n, uint32_t channels, const uint8_t *__restrict__ lo,
                                                                   const uint8_t *__restrict__ hi, const float *__restrict__ mins,
                                                                   const float *__restrict__ maxs, double levels, uint32_t shift,
                                                                   float *__restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * GS_BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x; i < n; i += stride) {
        const uint32_t c = (uint32_t)(i % channels);
        const uint32_t q = hi != nullptr ? (((uint32_t)hi[i] << 8) + (uint32_t)lo[i]) : ((uint32_t)lo[i] >> shift);
        const double norm = (double)q / levels;
        const float range = __fsub_rn(maxs[c], mins[c]);
        out[i] = (float)(norm * (double)range + (double)mins[c]);
    }
}

} // namespace

#define GS_ACT_DISPATCH(act, CALL)                                              \
    switch (act) {                                                             \
        case GS_ACT_NONE: { constexpr int A = GS_ACT_NONE; CALL; } break;      \
        case GS_ACT_EXP: { constexpr int A = GS_ACT_EXP; CALL; } break;        \
        default: { constexpr int A = GS_ACT_SIGMOID; CALL; } break;            \
    }

extern "C" int32_t gs_quantize_noise_fwd(
    uint64_t n, const float *x, const float *noise, float lo, float hi, float q_step, int32_t activation, float *out,
    gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(x && noise && out, "null pointer");
    GS_CHECK_ARG(activation >= GS_ACT_NONE && activation <= GS_ACT_SIGMOID, "unknown activation");
    const int vec = (int)aligned16(x, noise, out);
    GS_ACT_DISPATCH(activation, hipLaunchKernelGGL(quant_noise_fwd_kernel<A>, dim3(stream_grid(n)), dim3(GS_BLOCK), 0, (hipStream_t)stream,
                                                   n, x, noise, lo, hi, q_step, out, vec));
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_quantize_noise_bwd(
    uint64_t n, const float *x, const float *v_out, float lo, float hi, int32_t activation, const float *out, float *v_x,
    gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(x && v_out && v_x, "null pointer");
    GS_CHECK_ARG(activation >= GS_ACT_NONE && activation <= GS_ACT_SIGMOID, "unknown activation");
    GS_CHECK_ARG(activation == GS_ACT_NONE || out != nullptr, "an activation's gradient needs the forward output");
    const int vec = (int)(aligned16(x, v_out, v_x) && (out == nullptr || (uintptr_t)out % 16 == 0));
    GS_ACT_DISPATCH(activation, hipLaunchKernelGGL((quant_noise_bwd_kernel<A, true>), dim3(stream_grid(n)), dim3(GS_BLOCK), 0,
                                                   (hipStream_t)stream, n, x, v_out, out, lo, hi, v_x, vec));
    GS_CHECK_LAUNCH();
    return 0;
}

// ---------------------------------------------------------------------------
// Multi-tensor "noise" quantizer with the noise generated IN the kernel (BASELINE config 3: the hooks of four attributes in
// front of every render).  The reference draws the noise of every attribute with torch.empty_like(x).uniform_(-0.5, 0.5) from
// the device's default generator; torch's kernel for it is a Philox4x32-10 counter scheme
// (ATen/native/cuda/DistributionTemplates.h: distribution_elementwise_grid_stride_kernel): a grid of G = min(ceil(n / 256),
// CUs * (max threads per CU / 256)) workgroups of 256 threads, thread idx owns the Philox subsequence idx of (seed, offset) and
// its k-th draw of four 32-bit values serves the elements idx + (4 k + j) * 256 G, j = 0..3; u = 2^-32 v + 2^-32 in fp32,
// noise = u * (to - from) + from with the value `to` mapped back to `from`; the generator's offset then advances by
// 4 * (floor((n - 1) / (1024 G)) + 1).  The same draws are evaluated here, per (tensor, four consecutive idx, k) -- sixteen
// elements as four 16-byte accesses a stride apart -- so the outputs are bit-identical to uniform_ + gs_quantize_noise_fwd while the
// noise never exists in memory (12 -> 8 bytes per quantized float, 8 launches -> 1 for four attributes) and the RNG stream of
// the process stays exactly the reference's (the caller advances the generator by the same amounts).
// ---------------------------------------------------------------------------
struct QuantMultiArgs {
    gs_quant_desc d[GS_QUANT_MULTI_MAX];
    uint64_t item_end[GS_QUANT_MULTI_MAX]; // inclusive prefix sum of the tensors' work items ((idx, k) pairs)
    uint32_t stride[GS_QUANT_MULTI_MAX];   // 256 * G of the tensor
    uint32_t vec[GS_QUANT_MULTI_MAX];      // 16-byte accesses allowed
    uint32_t n;
    uint32_t seed_lo, seed_hi;
    uint32_t no_mask;                      // backward: gradients pass everywhere (the round mode, ops.py:73-75)
    float range[GS_QUANT_MULTI_MAX], q_norm[GS_QUANT_MULTI_MAX]; // round mode: hi - lo and 1 / (2^bits - 1) as torch rounds them
};

GS_DEV void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint32_t hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

GS_DEV float philox_noise(uint32_t v) { // torch's uniform_(-0.5, 0.5) of one 32-bit draw
    const float u = __fadd_rn(__fmul_rn((float)v, 2.3283064e-10f), 2.3283064e-10f); // (0, 1]
    const float val = __fadd_rn(__fmul_rn(u, 1.0f), -0.5f);
    return val == 0.5f ? -0.5f : val;
}

template <int ACT_DYN>
GS_DEV float q_act_dyn(float q, int act) { return act == GS_ACT_EXP ? q_act<GS_ACT_EXP>(q) : act == GS_ACT_SIGMOID ? q_act<GS_ACT_SIGMOID>(q) : q; }
GS_DEV float q_act_grad_dyn(float o, float g, int act) {
    return act == GS_ACT_EXP ? q_act_grad<GS_ACT_EXP>(o, g) : act == GS_ACT_SIGMOID ? q_act_grad<GS_ACT_SIGMOID>(o, g) : g;
}

// Forward work item: (tensor, idx4, k) = the four Philox subsequences 4 idx4 .. 4 idx4 + 3 at draw k -> 16 elements, as four
// 16-byte accesses a stride apart (element 4 idx4 + i + (4 k + j) stride uses value j of subsequence 4 idx4 + i).
// item_end counts these items (stride / 4 per draw).  vec[t] = 0: the tensor's pointers are not 16-byte aligned -> 4-byte accesses.
__global__ void __launch_bounds__(GS_BLOCK) quant_noise_multi_fwd_kernel(QuantMultiArgs a) {
    const uint64_t total = a.item_end[a.n - 1];
    const uint64_t gstride = (uint64_t)gridDim.x * GS_BLOCK;
    for (uint64_t it = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x; it < total; it += gstride) {
        uint32_t t = 0;
        while (it >= a.item_end[t]) ++t; // (<= 8 tensors)
        const gs_quant_desc d = a.d[t];
        const uint64_t local = it - (t ? a.item_end[t - 1] : 0);
        const uint32_t stride = a.stride[t], q4 = stride / 4;
        const uint32_t idx0 = 4u * (uint32_t)(local % q4);
        const uint64_t k = local / q4;
        const uint64_t e0 = (uint64_t)idx0 + 4ull * k * stride;
        if (e0 >= d.n) continue;
        const uint64_t ctr = d.philox_offset / 4 + k;
        uint32_t r[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) philox4x32_10((uint32_t)ctr, (uint32_t)(ctr >> 32), idx0 + (uint32_t)i, 0u, a.seed_lo, a.seed_hi, r[i]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const uint64_t e = e0 + (uint64_t)j * stride;
            if (e >= d.n) break;
            if (a.vec[t] && e + 4 <= d.n) {
                const float4 x = *reinterpret_cast<const float4 *>(d.x + e);
                float4 o;
                o.x = q_act_dyn<0>(q_noise(x.x, philox_noise(r[0][j]), d.lo, d.hi, d.q_step), d.activation);
                o.y = q_act_dyn<0>(q_noise(x.y, philox_noise(r[1][j]), d.lo, d.hi, d.q_step), d.activation);
                o.z = q_act_dyn<0>(q_noise(x.z, philox_noise(r[2][j]), d.lo, d.hi, d.q_step), d.activation);
                o.w = q_act_dyn<0>(q_noise(x.w, philox_noise(r[3][j]), d.lo, d.hi, d.q_step), d.activation);
                *reinterpret_cast<float4 *>(d.out + e) = o;
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (e + i < d.n) d.out[e + i] = q_act_dyn<0>(q_noise(d.x[e + i], philox_noise(r[i][j]), d.lo, d.hi, d.q_step), d.activation);
            }
        }
    }
}

// Backward work item: four consecutive elements of a tensor (no noise involved: plain streaming).
__global__ void __launch_bounds__(GS_BLOCK) quant_noise_multi_bwd_kernel(QuantMultiArgs a) {
    const uint64_t total = a.item_end[a.n - 1];
    const uint64_t gstride = (uint64_t)gridDim.x * GS_BLOCK;
    for (uint64_t it = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x; it < total; it += gstride) {
        uint32_t t = 0;
        while (it >= a.item_end[t]) ++t;
        const gs_quant_desc d = a.d[t];
        const uint64_t e = 4ull * (it - (t ? a.item_end[t - 1] : 0));
        if (a.vec[t] && e + 4 <= d.n) {
            const float4 x = *reinterpret_cast<const float4 *>(d.x + e);
            float4 g = *reinterpret_cast<const float4 *>(d.v_out + e);
            if (d.activation != GS_ACT_NONE) {
                const float4 o = *reinterpret_cast<const float4 *>(d.out + e);
                g.x = q_act_grad_dyn(o.x, g.x, d.activation); g.y = q_act_grad_dyn(o.y, g.y, d.activation);
                g.z = q_act_grad_dyn(o.z, g.z, d.activation); g.w = q_act_grad_dyn(o.w, g.w, d.activation);
            }
            float4 r = g;
            if (!a.no_mask) {
                r.x = q_mask(x.x, g.x, d.lo, d.hi); r.y = q_mask(x.y, g.y, d.lo, d.hi);
                r.z = q_mask(x.z, g.z, d.lo, d.hi); r.w = q_mask(x.w, g.w, d.lo, d.hi);
            }
            *reinterpret_cast<float4 *>(d.v_x + e) = r;
        } else {
            for (uint64_t i = e; i < e + 4 && i < d.n; ++i) {
                float g = d.v_out[i];
                if (d.activation != GS_ACT_NONE) g = q_act_grad_dyn(d.out[i], g, d.activation);
                d.v_x[i] = a.no_mask ? g : q_mask(d.x[i], g, d.lo, d.hi);
            }
        }
    }
}

// Round mode, all hooked tensors of a step in one launch (work item: four consecutive elements of a tensor): the arithmetic of
// quant_round_fwd_kernel -- the parameter (d.v_x, the writable alias of d.x) is clamped IN PLACE (ops.py:63), out = the grid value.
__global__ void __launch_bounds__(GS_BLOCK) quant_round_multi_fwd_kernel(QuantMultiArgs a) {
    const uint64_t total = a.item_end[a.n - 1];
    const uint64_t gstride = (uint64_t)gridDim.x * GS_BLOCK;
    for (uint64_t it = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x; it < total; it += gstride) {
        uint32_t t = 0;
        while (it >= a.item_end[t]) ++t;
        const gs_quant_desc d = a.d[t];
        const float range = a.range[t], qn = a.q_norm[t];
        const uint64_t e = 4ull * (it - (t ? a.item_end[t - 1] : 0));
        if (a.vec[t] && e + 4 <= d.n) {
            const float4 x0 = *reinterpret_cast<const float4 *>(d.x + e);
            float4 x;
            x.x = q_clamp(x0.x, d.lo, d.hi); x.y = q_clamp(x0.y, d.lo, d.hi);
            x.z = q_clamp(x0.z, d.lo, d.hi); x.w = q_clamp(x0.w, d.lo, d.hi);
            if (x.x != x0.x || x.y != x0.y || x.z != x0.z || x.w != x0.w) *reinterpret_cast<float4 *>(d.v_x + e) = x; // (only where it clamps)
            float4 o;
            o.x = q_act_dyn<0>(q_round(x.x, d.lo, range, qn), d.activation); o.y = q_act_dyn<0>(q_round(x.y, d.lo, range, qn), d.activation);
            o.z = q_act_dyn<0>(q_round(x.z, d.lo, range, qn), d.activation); o.w = q_act_dyn<0>(q_round(x.w, d.lo, range, qn), d.activation);
            *reinterpret_cast<float4 *>(d.out + e) = o;
        } else {
            for (uint64_t i = e; i < e + 4 && i < d.n; ++i) {
                const float c0 = d.x[i], c = q_clamp(c0, d.lo, d.hi);
                if (c != c0) d.v_x[i] = c;
                d.out[i] = q_act_dyn<0>(q_round(c, d.lo, range, qn), d.activation);
            }
        }
    }
}

static int32_t quant_multi_launch(uint32_t n_tensors, const gs_quant_desc *descs, uint64_t seed, uint32_t grid_cap, bool bwd,
                                  hipStream_t st, const char *who, const float *round_range = nullptr, const float *round_q_norm = nullptr,
                                  bool no_mask = false) {
    QuantMultiArgs a;
    uint64_t items = 0;
    a.n = 0;
    a.no_mask = no_mask ? 1u : 0u;
    const bool round_fwd = round_range != nullptr;
    for (uint32_t t = 0; t < n_tensors; ++t) {
        const gs_quant_desc &d = descs[t];
        if (d.n == 0) continue;
        if (round_fwd && (d.v_x == nullptr || (const float *)d.v_x != d.x)) {
            gs_set_error("%s: descriptor %u: v_x must be the writable alias of x (the parameter is clamped in place)", who, t);
            return 1;
        }
        if (!d.x || (bwd ? (!d.v_out || !d.v_x) : !d.out) || d.activation < GS_ACT_NONE || d.activation > GS_ACT_SIGMOID ||
            (bwd && d.activation != GS_ACT_NONE && !d.out) || d.philox_offset % 4 != 0) {
            gs_set_error("%s: bad descriptor %u (null pointer, unknown activation or an offset that is not a multiple of 4)", who, t);
            return 1;
        }
        const uint64_t blocks = (d.n + GS_BLOCK - 1) / GS_BLOCK;
        const uint32_t G = (uint32_t)(blocks < grid_cap ? blocks : grid_cap);
        a.stride[a.n] = GS_BLOCK * G;
        if (round_fwd) {
            items += (d.n + 3) / 4;
            a.vec[a.n] = ((uintptr_t)d.x % 16 == 0 && (uintptr_t)d.out % 16 == 0) ? 1u : 0u;
            a.range[a.n] = round_range[t];
            a.q_norm[a.n] = round_q_norm[t];
        } else if (bwd) {
            items += (d.n + 3) / 4;
            a.vec[a.n] = ((uintptr_t)d.x % 16 == 0 && (uintptr_t)d.v_out % 16 == 0 && (uintptr_t)d.v_x % 16 == 0 && (uintptr_t)d.out % 16 == 0) ? 1u : 0u;
        } else {
            const uint64_t calls = (d.n - 1) / ((uint64_t)a.stride[a.n] * 4) + 1;
            items += (uint64_t)(a.stride[a.n] / 4) * calls;
            a.vec[a.n] = ((uintptr_t)d.x % 16 == 0 && (uintptr_t)d.out % 16 == 0) ? 1u : 0u;
        }
        a.item_end[a.n] = items;
        a.d[a.n] = d;
        ++a.n;
    }
    if (a.n == 0) return 0;
    a.seed_lo = (uint32_t)seed;
    a.seed_hi = (uint32_t)(seed >> 32);
    uint64_t blocks = (items + GS_BLOCK - 1) / GS_BLOCK;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (round_fwd) hipLaunchKernelGGL(quant_round_multi_fwd_kernel, dim3((uint32_t)blocks), dim3(GS_BLOCK), 0, st, a);
    else if (bwd) hipLaunchKernelGGL(quant_noise_multi_bwd_kernel, dim3((uint32_t)blocks), dim3(GS_BLOCK), 0, st, a);
    else hipLaunchKernelGGL(quant_noise_multi_fwd_kernel, dim3((uint32_t)blocks), dim3(GS_BLOCK), 0, st, a);
    return 0;
}

extern "C" uint64_t gs_quantize_philox_advance(uint64_t n, uint32_t grid_cap) {
    if (n == 0 || grid_cap == 0) return 0;
    const uint64_t blocks = (n + GS_BLOCK - 1) / GS_BLOCK;
    const uint64_t G = blocks < grid_cap ? blocks : grid_cap;
    return ((n - 1) / (GS_BLOCK * G * 4) + 1) * 4;
}

extern "C" int32_t gs_quantize_noise_multi_fwd(uint32_t n_tensors, const gs_quant_desc *descs, uint64_t philox_seed, uint32_t grid_cap,
                                               gs_stream_t stream) {
    GS_CHECK_ARG(n_tensors <= GS_QUANT_MULTI_MAX && (n_tensors == 0 || descs != nullptr), "up to GS_QUANT_MULTI_MAX descriptors");
    GS_CHECK_ARG(grid_cap > 0, "grid_cap (CUs * max threads per CU / 256 of the device, what torch sizes uniform_'s grid by) must be > 0");
    const int32_t rc = quant_multi_launch(n_tensors, descs, philox_seed, grid_cap, false, (hipStream_t)stream, "gs_quantize_noise_multi_fwd");
    if (rc) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_quantize_noise_multi_bwd(uint32_t n_tensors, const gs_quant_desc *descs, gs_stream_t stream) {
    GS_CHECK_ARG(n_tensors <= GS_QUANT_MULTI_MAX && (n_tensors == 0 || descs != nullptr), "up to GS_QUANT_MULTI_MAX descriptors");
    const int32_t rc = quant_multi_launch(n_tensors, descs, 0, 1u << 20, true, (hipStream_t)stream, "gs_quantize_noise_multi_bwd");
    if (rc) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_quantize_round_multi_fwd(uint32_t n_tensors, const gs_quant_desc *descs, const float *ranges, const float *q_step_norms,
                                               gs_stream_t stream) {
    GS_CHECK_ARG(n_tensors <= GS_QUANT_MULTI_MAX && (n_tensors == 0 || (descs != nullptr && ranges != nullptr && q_step_norms != nullptr)),
                 "up to GS_QUANT_MULTI_MAX descriptors with their range / step tables");
    const int32_t rc = quant_multi_launch(n_tensors, descs, 0, 1u << 20, false, (hipStream_t)stream, "gs_quantize_round_multi_fwd", ranges, q_step_norms);
    if (rc) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}

// (the round mode's backward is the identity; with activations fused in: v_x = v_out x the activation's derivative, no clamp mask)
extern "C" int32_t gs_quantize_round_multi_bwd(uint32_t n_tensors, const gs_quant_desc *descs, gs_stream_t stream) {
    GS_CHECK_ARG(n_tensors <= GS_QUANT_MULTI_MAX && (n_tensors == 0 || descs != nullptr), "up to GS_QUANT_MULTI_MAX descriptors");
    const int32_t rc = quant_multi_launch(n_tensors, descs, 0, 1u << 20, true, (hipStream_t)stream, "gs_quantize_round_multi_bwd", nullptr, nullptr, true);
    if (rc) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_quantize_round_fwd(
    uint64_t n, float *x_inplace, float lo, float hi, float range, float q_step_norm, int32_t activation, float *out,
    gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(x_inplace && out, "null pointer");
    GS_CHECK_ARG(activation >= GS_ACT_NONE && activation <= GS_ACT_SIGMOID, "unknown activation");
    const int vec = (int)aligned16(x_inplace, out, out);
    GS_ACT_DISPATCH(activation, hipLaunchKernelGGL(quant_round_fwd_kernel<A>, dim3(stream_grid(n)), dim3(GS_BLOCK), 0, (hipStream_t)stream,
                                                   n, x_inplace, lo, hi, range, q_step_norm, out, vec));
    GS_CHECK_LAUNCH();
    return 0;
}

// the round mode's backward is the identity (ops.py:73-75) -- no kernel -- unless an activation was fused in
extern "C" int32_t gs_quantize_round_bwd(uint64_t n, const float *v_out, int32_t activation, const float *out, float *v_x,
                                         gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(v_out && v_x && out, "null pointer");
    GS_CHECK_ARG(activation > GS_ACT_NONE && activation <= GS_ACT_SIGMOID, "the identity gradient needs no kernel");
    const int vec = (int)aligned16(out, v_out, v_x);
    GS_ACT_DISPATCH(activation, hipLaunchKernelGGL((quant_noise_bwd_kernel<A, false>), dim3(stream_grid(n)), dim3(GS_BLOCK), 0,
                                                   (hipStream_t)stream, n, (const float *)nullptr, v_out, out, 0.f, 0.f, v_x, vec));
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_grid_quantize(uint64_t n, uint32_t channels, const float *x, const float *mins, const float *maxs,
                                    uint32_t bits, uint8_t *plane_lo, uint8_t *plane_hi, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(x && mins && maxs && plane_lo, "null pointer");
    GS_CHECK_ARG(channels >= 1 && n % channels == 0, "n must be a multiple of channels");
    GS_CHECK_ARG((bits >= 1 && bits <= 8 && plane_hi == nullptr) || (bits == 16 && plane_hi != nullptr),
                 "bits must be 1..8 (one plane) or 16 (two planes)");
    const float levels = (float)((1u << bits) - 1u);
    const uint32_t shift = bits <= 8 ? 8u - bits : 0u;
    const uint32_t blocks = (uint32_t)(gs_div_up(n, GS_BLOCK) < 8192u ? gs_div_up(n, GS_BLOCK) : 8192u);
    hipLaunchKernelGGL(grid_quantize_kernel, dim3(blocks), dim3(GS_BLOCK), 0, (hipStream_t)stream, n, channels, x, mins, maxs, levels,
                       shift, plane_lo, plane_hi);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_grid_dequantize(uint64_t n, uint32_t channels, const uint8_t *plane_lo, const uint8_t *plane_hi,
                                      const float *mins, const float *maxs, uint32_t bits, float *out, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(plane_lo && mins && maxs && out, "null pointer");
    GS_CHECK_ARG(channels >= 1 && n % channels == 0, "n must be a multiple of channels");
    GS_CHECK_ARG((bits >= 1 && bits <= 8 && plane_hi == nullptr) || (bits == 16 && plane_hi != nullptr),
                 "bits must be 1..8 (one plane) or 16 (two planes)");
    const double levels = (double)((1u << bits) - 1u);
    const uint32_t shift = bits <= 8 ? 8u - bits : 0u;
    const uint32_t blocks = (uint32_t)(gs_div_up(n, GS_BLOCK) < 8192u ? gs_div_up(n, GS_BLOCK) : 8192u);
    hipLaunchKernelGGL(grid_dequantize_kernel, dim3(blocks), dim3(GS_BLOCK), 0, (hipStream_t)stream, n, channels, plane_lo, plane_hi,
                       mins, maxs, levels, shift, out);
    GS_CHECK_LAUNCH();
    return 0;
}
