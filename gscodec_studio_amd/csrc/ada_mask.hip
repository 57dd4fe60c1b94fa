// ada_mask.hip -- Q3: the learnable per-splat mask on the higher SH bands ("shN adaptive mask") of the compression-simulation
// hooks, and the gradient-threshold variant (gfx950).
//
// Replaces the elementwise torch chains of
//   gsplat/compression_simulation/ada_mask.py:32-40   AnnealingMask.forward     out = x * sigmoid(logit / T)   (training)
//                                                                                out = x * (sigmoid(logit) >= 0.5)  (eval)
//   gsplat/compression_simulation/ada_mask.py:46-58   get_sparsity_loss: mean(sigmoid(logit / T)) (the rest is scalar math)
//   gsplat/compression_simulation/ada_mask.py:60-62   get_mask_ratio
//   gsplat/compression_simulation/simulation.py:327-348  shN_gradient_threshold (the "gradient" strategy)
// x is the shN parameter [N, K-1, 3] (row = 3 (K-1) floats per splat, 45 at degree 3), logit one float per splat.
//
// Mapping: one 256-thread workgroup per block of 256 splats.  The block's slice of x is CONTIGUOUS (256 * row floats, a
// multiple of 16 bytes), so it streams as 16-byte loads / stores whatever `row` is; the 256 masks are evaluated once (one
// sigmoid per thread) and handed over through LDS; an element's splat is `e / row` by a multiply-high.  The backward's
// per-splat dot product sum_j v_out[n,j] x[n,j] (the mask's gradient) is reduced deterministically: the products are staged
// through LDS in element order and thread s adds up row s -- no atomics, no cross-lane traffic, the same sum every run.
// HBM: forward 8 row + 4 bytes per splat, backward 12 row + 8.
//
// Arithmetic: IEEE fp32, no contraction (file compiled with -ffp-contract=off), torch's CPU operation order -- the division
// logit / T is a true division, sigmoid(v) = 1 / (1 + exp(-v)); given the same mask value the product x * mask is bit-exact.
#include "gs_common.h"

namespace {

constexpr int AM_SPLATS = 256; // splats per workgroup

GS_DEV float am_sigmoid(float v) { return gs_mask_sigmoid(v); }
GS_DEV float am_mask(float logit, float temperature, int binary) { return gs_mask_value(logit, temperature, binary); } // ada_mask.py:37, 39, 44

// e / row for e < 2^26, row < 2^6 .. 2^10: floor(e * ceil(2^32 / row) / 2^32) is exact while e * row < 2^32
GS_DEV uint32_t am_div(uint32_t e, uint32_t magic) { return __umulhi(e, magic); }

__global__ void __launch_bounds__(AM_SPLATS) shn_mask_fwd_kernel(uint64_t n, uint32_t row, uint32_t magic, const float *__restrict__ x,
                                                                 const float *__restrict__ logits, float temperature, int binary,
                                                                 float *__restrict__ out, int vec_ok) {
    __shared__ float s_mask[AM_SPLATS];
    const uint64_t s0 = (uint64_t)blockIdx.x * AM_SPLATS;
    const uint32_t cnt = (uint32_t)(n - s0 < AM_SPLATS ? n - s0 : AM_SPLATS);
    const uint32_t t = threadIdx.x;
    if (t < cnt) s_mask[t] = am_mask(logits[s0 + t], temperature, binary);
    __syncthreads();
    const uint64_t base = s0 * row;
    const uint32_t n_el = cnt * row;
    const float *xb = x + base;
    float *ob = out + base;
    const uint32_t nv = vec_ok ? n_el / 4 : 0;
    for (uint32_t i = t; i < nv; i += AM_SPLATS) {
        const float4 a = reinterpret_cast<const float4 *>(xb)[i];
        const uint32_t e = 4 * i;
        float4 r;
        r.x = __fmul_rn(a.x, s_mask[am_div(e, magic)]);
        r.y = __fmul_rn(a.y, s_mask[am_div(e + 1, magic)]);
        r.z = __fmul_rn(a.z, s_mask[am_div(e + 2, magic)]);
        r.w = __fmul_rn(a.w, s_mask[am_div(e + 3, magic)]);
        reinterpret_cast<float4 *>(ob)[i] = r;
    }
    for (uint32_t e = nv * 4 + t; e < n_el; e += AM_SPLATS) ob[e] = __fmul_rn(xb[e], s_mask[am_div(e, magic)]);
}

// v_x = v_out * mask;  v_logits[n] = (sum_j v_out[n,j] x[n,j]) * mask (1 - mask) / T   (autograd of ada_mask.py:37, 40:
// mul backward, sum over the broadcast dims, sigmoid_backward = g (1 - y) y, div backward = g / T).
// LDS: row * 256 floats of products (46 KB at degree 3); dynamic.
__global__ void __launch_bounds__(AM_SPLATS) shn_mask_bwd_kernel(uint64_t n, uint32_t row, uint32_t magic, const float *__restrict__ x,
                                                                 const float *__restrict__ logits, float temperature, int binary,
                                                                 const float *__restrict__ v_out, float *__restrict__ v_x,
                                                                 float *__restrict__ v_logits, int vec_ok) {
    extern __shared__ float s_dyn[];
    float *s_mask = s_dyn;            // [256]
    float *s_prod = s_dyn + AM_SPLATS; // [256 * row]
    const uint64_t s0 = (uint64_t)blockIdx.x * AM_SPLATS;
    const uint32_t cnt = (uint32_t)(n - s0 < AM_SPLATS ? n - s0 : AM_SPLATS);
    const uint32_t t = threadIdx.x;
    float my_mask = 0.f;
    if (t < cnt) {
        my_mask = am_mask(logits[s0 + t], temperature, binary);
        s_mask[t] = my_mask;
    }
    __syncthreads();
    const uint64_t base = s0 * row;
    const uint32_t n_el = cnt * row;
    const float *xb = x + base;
    const float *gb = v_out + base;
    float *ob = v_x ? v_x + base : nullptr;
    const bool want_dot = v_logits != nullptr;
    const uint32_t nv = vec_ok ? n_el / 4 : 0;
    for (uint32_t i = t; i < nv; i += AM_SPLATS) {
        const float4 g = reinterpret_cast<const float4 *>(gb)[i];
        const uint32_t e = 4 * i;
        if (ob) {
            float4 r;
            r.x = __fmul_rn(g.x, s_mask[am_div(e, magic)]);
            r.y = __fmul_rn(g.y, s_mask[am_div(e + 1, magic)]);
            r.z = __fmul_rn(g.z, s_mask[am_div(e + 2, magic)]);
            r.w = __fmul_rn(g.w, s_mask[am_div(e + 3, magic)]);
            reinterpret_cast<float4 *>(ob)[i] = r;
        }
        if (want_dot) {
            const float4 a = reinterpret_cast<const float4 *>(xb)[i];
            float4 p;
            p.x = __fmul_rn(g.x, a.x); p.y = __fmul_rn(g.y, a.y); p.z = __fmul_rn(g.z, a.z); p.w = __fmul_rn(g.w, a.w);
            reinterpret_cast<float4 *>(s_prod)[i] = p;
        }
    }
    for (uint32_t e = nv * 4 + t; e < n_el; e += AM_SPLATS) {
        const float g = gb[e];
        if (ob) ob[e] = __fmul_rn(g, s_mask[am_div(e, magic)]);
        if (want_dot) s_prod[e] = __fmul_rn(g, xb[e]);
    }
    if (!want_dot) return;
    __syncthreads();
    if (t < cnt) {
        // row t of the staged products: consecutive threads are `row` words apart -- conflict-free for odd rows (45 at degree
        // 3, 9 / 24 / 72 at degrees 1 / 2 / 4: the even ones pay a few-way conflict on a pass that is not the bottleneck)
        const float *p = s_prod + t * row;
        float acc = 0.f;
        for (uint32_t j = 0; j < row; ++j) acc = __fadd_rn(acc, p[j]);
        const float v_mask = acc;
        const float v_sig = __fmul_rn(__fmul_rn(v_mask, __fsub_rn(1.f, my_mask)), my_mask); // sigmoid_backward: g (1 - y) y
        v_logits[s0 + t] = __fdiv_rn(v_sig, temperature);
    }
}

__global__ void __launch_bounds__(GS_BLOCK) mask_values_kernel(uint64_t n, const float *__restrict__ logits, float temperature, int binary,
                                                               float *__restrict__ out) {
    const uint64_t stride = (uint64_t)gridDim.x * GS_BLOCK;
    for (uint64_t i = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x; i < n; i += stride) out[i] = am_mask(logits[i], temperature, binary);
}

// sum_i sigmoid(logit_i / T) (binary: the count of sigmoid(logit_i) >= 0.5): per-workgroup partial sums in double, in a fixed
// order; the finishing kernel adds the partials up in index order -- deterministic.
constexpr int AM_RED_BLOCKS = 512;

__global__ void __launch_bounds__(GS_BLOCK) mask_sum_partial_kernel(uint64_t n, const float *__restrict__ logits, float temperature,
                                                                    int binary, double *__restrict__ partials) {
    __shared__ double s_red[GS_BLOCK];
    double acc = 0.0;
    // contiguous slice per workgroup, strided by thread inside it: fixed assignment -> fixed summation order
    const uint64_t per = (n + gridDim.x - 1) / gridDim.x;
    const uint64_t lo = (uint64_t)blockIdx.x * per;
    const uint64_t hi = lo + per < n ? lo + per : n;
    for (uint64_t i = lo + threadIdx.x; i < hi; i += GS_BLOCK) acc += (double)am_mask(logits[i], temperature, binary);
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = GS_BLOCK / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) s_red[threadIdx.x] += s_red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) partials[blockIdx.x] = s_red[0];
}

__global__ void __launch_bounds__(64) mask_sum_finish_kernel(uint32_t n_partials, const double *__restrict__ partials, float divisor,
                                                             float *__restrict__ out) {
    if (threadIdx.x != 0) return;
    double acc = 0.0;
    for (uint32_t i = 0; i < n_partials; ++i) acc += partials[i];
    out[0] = __fdiv_rn((float)acc, divisor); // torch.mean / `count / N`: an fp32 division of the (here: correctly rounded) sum
}

// v_logits[i] = (g[0] / divisor) * (1 - y) y / T with y = sigmoid(logit_i / T): the gradient of sum_i y_i / divisor
// (mean backward, sigmoid_backward, div backward in torch's order)
__global__ void __launch_bounds__(GS_BLOCK) mask_mean_bwd_kernel(uint64_t n, const float *__restrict__ logits, float temperature,
                                                                 const float *__restrict__ v_mean, float divisor,
                                                                 float *__restrict__ v_logits) {
    const uint64_t stride = (uint64_t)gridDim.x * GS_BLOCK;
    const float g = __fdiv_rn(v_mean[0], divisor);
    for (uint64_t i = (uint64_t)blockIdx.x * GS_BLOCK + threadIdx.x; i < n; i += stride) {
        const float y = am_sigmoid(__fdiv_rn(logits[i], temperature));
        v_logits[i] = __fdiv_rn(__fmul_rn(__fmul_rn(g, __fsub_rn(1.f, y)), y), temperature);
    }
}

// "gradient" strategy, pass 1: flags[n] = 1 where every value of row n is exactly 0 (simulation.py:331); *n_zero += count
__global__ void __launch_bounds__(AM_SPLATS) shn_zero_rows_kernel(uint64_t n, uint32_t row, uint32_t magic, const float *__restrict__ x,
                                                                  uint8_t *__restrict__ flags, unsigned long long *__restrict__ n_zero,
                                                                  int vec_ok) {
    __shared__ uint32_t s_nonzero[AM_SPLATS];
    __shared__ uint32_t s_count;
    const uint64_t s0 = (uint64_t)blockIdx.x * AM_SPLATS;
    const uint32_t cnt = (uint32_t)(n - s0 < AM_SPLATS ? n - s0 : AM_SPLATS);
    const uint32_t t = threadIdx.x;
    s_nonzero[t] = 0;
    if (t == 0) s_count = 0;
    __syncthreads();
    const float *xb = x + s0 * row;
    const uint32_t n_el = cnt * row;
    const uint32_t nv = vec_ok ? n_el / 4 : 0;
    // (x == 0) is true for +0 and -0 and false for NaN, like torch's `param_value == 0`
    for (uint32_t i = t; i < nv; i += AM_SPLATS) {
        const float4 a = reinterpret_cast<const float4 *>(xb)[i];
        const uint32_t e = 4 * i;
        if (!(a.x == 0.f)) s_nonzero[am_div(e, magic)] = 1;
        if (!(a.y == 0.f)) s_nonzero[am_div(e + 1, magic)] = 1;
        if (!(a.z == 0.f)) s_nonzero[am_div(e + 2, magic)] = 1;
        if (!(a.w == 0.f)) s_nonzero[am_div(e + 3, magic)] = 1;
    }
    for (uint32_t e = nv * 4 + t; e < n_el; e += AM_SPLATS)
        if (!(xb[e] == 0.f)) s_nonzero[am_div(e, magic)] = 1;
    __syncthreads();
    if (t < cnt) {
        const uint32_t z = s_nonzero[t] ? 0u : 1u;
        flags[s0 + t] = (uint8_t)z;
        if (z) atomicAdd(&s_count, 1u);
    }
    __syncthreads();
    if (t == 0 && s_count) atomicAdd(n_zero, (unsigned long long)s_count);
}

// pass 2 (simulation.py:333-348): threshold = 2e-3 when fewer than 10 % of the splats have a non-zero row, else 100; the
// gradient rows of splats whose row is all zero AND whose gradient's Frobenius norm is below the threshold are zeroed in place.
__global__ void __launch_bounds__(AM_SPLATS) shn_grad_threshold_kernel(uint64_t n, uint32_t row, uint32_t magic,
                                                                       const uint8_t *__restrict__ flags,
                                                                       const unsigned long long *__restrict__ n_zero,
                                                                       float *__restrict__ grad, int vec_ok) {
    extern __shared__ float s_dyn[];
    float *s_sq = s_dyn;                                   // [256 * row] squares, element order
    uint32_t *s_kill = reinterpret_cast<uint32_t *>(s_dyn + (size_t)AM_SPLATS * row); // [256]
    const uint64_t s0 = (uint64_t)blockIdx.x * AM_SPLATS;
    const uint32_t cnt = (uint32_t)(n - s0 < AM_SPLATS ? n - s0 : AM_SPLATS);
    const uint32_t t = threadIdx.x;
    // torch: 1 - zero_mask.sum() / N in fp32; `< 0.10` against the fp32 tensor
    const float ratio = __fsub_rn(1.f, __fdiv_rn((float)(long long)*n_zero, (float)(long long)n));
    const float thr = ratio < 0.10f ? 2e-3f : 100.f;
    // a workgroup without a zero row has nothing to do (the common case until the mask bites)
    uint32_t mine = t < cnt ? flags[s0 + t] : 0u;
    if (!__syncthreads_or((int)mine)) return;
    float *gb = grad + s0 * row;
    const uint32_t n_el = cnt * row;
    const uint32_t nv = vec_ok ? n_el / 4 : 0;
    for (uint32_t i = t; i < nv; i += AM_SPLATS) {
        const float4 g = reinterpret_cast<const float4 *>(gb)[i];
        float4 p;
        p.x = __fmul_rn(g.x, g.x); p.y = __fmul_rn(g.y, g.y); p.z = __fmul_rn(g.z, g.z); p.w = __fmul_rn(g.w, g.w);
        reinterpret_cast<float4 *>(s_sq)[i] = p;
    }
    for (uint32_t e = nv * 4 + t; e < n_el; e += AM_SPLATS) s_sq[e] = __fmul_rn(gb[e], gb[e]);
    __syncthreads();
    uint32_t kill = 0;
    if (mine) {
        const float *p = s_sq + t * row;
        float acc = 0.f;
        for (uint32_t j = 0; j < row; ++j) acc = __fadd_rn(acc, p[j]);
        kill = sqrtf(acc) < thr ? 1u : 0u;
    }
    s_kill[t] = kill;
    if (!__syncthreads_or((int)kill)) return;
    for (uint32_t e = t; e < n_el; e += AM_SPLATS)
        if (s_kill[am_div(e, magic)]) gb[e] = 0.f;
}

bool am_args_ok(uint64_t n, uint32_t row) { return row >= 1 && row <= 4096 && n < (1ull << 40); }
uint32_t am_magic(uint32_t row) { return (uint32_t)((0x100000000ull + row - 1) / row); } // ceil(2^32 / row); row = 1 wraps to 0
bool am_al16(const void *p) { return p == nullptr || (uintptr_t)p % 16 == 0; }

} // namespace

extern "C" int32_t gs_shn_mask_fwd(uint64_t n, uint32_t row, const float *x, const float *mask_logits, float temperature,
                                   int32_t binary, float *out, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(x && mask_logits && out, "null pointer");
    GS_CHECK_ARG(am_args_ok(n, row) && row >= 2, "row (floats per splat) must be in 2..4096");
    GS_CHECK_ARG(binary || temperature > 0.f, "temperature must be positive");
    const int vec = (int)(am_al16(x) && am_al16(out));
    hipLaunchKernelGGL(shn_mask_fwd_kernel, dim3(gs_div_up(n, AM_SPLATS)), dim3(AM_SPLATS), 0, (hipStream_t)stream, n, row, am_magic(row), x,
                       mask_logits, temperature, (int)(binary != 0), out, vec);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_shn_mask_bwd(uint64_t n, uint32_t row, const float *x, const float *mask_logits, float temperature,
                                   int32_t binary, const float *v_out, float *v_x, float *v_mask_logits, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(mask_logits && v_out, "null pointer");
    GS_CHECK_ARG(v_x || v_mask_logits, "nothing to compute");
    GS_CHECK_ARG(v_mask_logits == nullptr || x != nullptr, "the mask's gradient needs x");
    GS_CHECK_ARG(!(binary && v_mask_logits), "the binary (eval) mask has no gradient");
    GS_CHECK_ARG(am_args_ok(n, row) && row >= 2, "row (floats per splat) must be in 2..4096");
    GS_CHECK_ARG(binary || temperature > 0.f, "temperature must be positive");
    const size_t lds = sizeof(float) * (AM_SPLATS + (v_mask_logits ? (size_t)AM_SPLATS * row : 0));
    GS_CHECK_ARG(lds <= 160 * 1024, "row too long for the LDS-staged reduction (max 159 floats per splat)");
    if (lds > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(shn_mask_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                           (int)lds);
        GS_CHECK_ARG(e == hipSuccess, "cannot raise the dynamic LDS limit");
    }
    const int vec = (int)(am_al16(x) && am_al16(v_out) && am_al16(v_x));
    hipLaunchKernelGGL(shn_mask_bwd_kernel, dim3(gs_div_up(n, AM_SPLATS)), dim3(AM_SPLATS), lds, (hipStream_t)stream, n, row, am_magic(row), x,
                       mask_logits, temperature, (int)(binary != 0), v_out, v_x, v_mask_logits, vec);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_mask_values(uint64_t n, const float *mask_logits, float temperature, int32_t binary, float *out, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(mask_logits && out, "null pointer");
    GS_CHECK_ARG(binary || temperature > 0.f, "temperature must be positive");
    const uint32_t blocks = gs_div_up(n, GS_BLOCK) < 2048u ? gs_div_up(n, GS_BLOCK) : 2048u;
    hipLaunchKernelGGL(mask_values_kernel, dim3(blocks), dim3(GS_BLOCK), 0, (hipStream_t)stream, n, mask_logits, temperature,
                       (int)(binary != 0), out);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" size_t gs_mask_sum_temp_bytes(void) { return sizeof(double) * AM_RED_BLOCKS; }

extern "C" int32_t gs_mask_sum(uint64_t n, const float *mask_logits, float temperature, int32_t binary, float divisor, void *temp,
                               float *out, gs_stream_t stream) {
    GS_CHECK_ARG(out && temp, "null pointer");
    GS_CHECK_ARG(n == 0 || mask_logits, "null pointer");
    GS_CHECK_ARG(binary || temperature > 0.f, "temperature must be positive");
    const uint32_t blocks = n == 0 ? 0u : (gs_div_up(n, GS_BLOCK * 8) < (uint32_t)AM_RED_BLOCKS ? gs_div_up(n, GS_BLOCK * 8) : (uint32_t)AM_RED_BLOCKS);
    if (blocks)
        hipLaunchKernelGGL(mask_sum_partial_kernel, dim3(blocks), dim3(GS_BLOCK), 0, (hipStream_t)stream, n, mask_logits, temperature,
                           (int)(binary != 0), (double *)temp);
    hipLaunchKernelGGL(mask_sum_finish_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, blocks, (const double *)temp, divisor, out);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_mask_mean_bwd(uint64_t n, const float *mask_logits, float temperature, const float *v_mean, float divisor,
                                    float *v_mask_logits, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(mask_logits && v_mean && v_mask_logits, "null pointer");
    GS_CHECK_ARG(temperature > 0.f, "temperature must be positive");
    const uint32_t blocks = gs_div_up(n, GS_BLOCK) < 2048u ? gs_div_up(n, GS_BLOCK) : 2048u;
    hipLaunchKernelGGL(mask_mean_bwd_kernel, dim3(blocks), dim3(GS_BLOCK), 0, (hipStream_t)stream, n, mask_logits, temperature, v_mean, divisor,
                       v_mask_logits);
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_shn_grad_threshold(uint64_t n, uint32_t row, const float *x, float *grad_inplace, uint8_t *zero_rows,
                                         uint64_t *n_zero, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(x && grad_inplace && zero_rows && n_zero, "null pointer");
    GS_CHECK_ARG(am_args_ok(n, row) && row >= 2, "row (floats per splat) must be in 2..4096");
    const size_t lds = sizeof(float) * ((size_t)AM_SPLATS * row + AM_SPLATS);
    GS_CHECK_ARG(lds <= 160 * 1024, "row too long for the LDS-staged reduction (max 159 floats per splat)");
    hipError_t e = hipMemsetAsync(n_zero, 0, sizeof(uint64_t), (hipStream_t)stream);
    GS_CHECK_ARG(e == hipSuccess, "memset failed");
    hipLaunchKernelGGL(shn_zero_rows_kernel, dim3(gs_div_up(n, AM_SPLATS)), dim3(AM_SPLATS), 0, (hipStream_t)stream, n, row, am_magic(row), x,
                       zero_rows, (unsigned long long *)n_zero, (int)am_al16(x));
    if (lds > 64 * 1024) {
        e = hipFuncSetAttribute(reinterpret_cast<const void *>(shn_grad_threshold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        GS_CHECK_ARG(e == hipSuccess, "cannot raise the dynamic LDS limit");
    }
    hipLaunchKernelGGL(shn_grad_threshold_kernel, dim3(gs_div_up(n, AM_SPLATS)), dim3(AM_SPLATS), lds, (hipStream_t)stream, n, row,
                       am_magic(row), zero_rows, (const unsigned long long *)n_zero, grad_inplace, (int)am_al16(grad_inplace));
    GS_CHECK_LAUNCH();
    return 0;
}
