// entropy.hip -- factorized-prior bits estimator (the rate term of the compression simulation),
// forward + backward, one fused kernel each way (gfx950).
//
// Replaces Entropy_factorized_optimized_refactor.forward of the reference
// (gsplat/compression_simulation/entropy_model.py:195-254): ~30 torch kernels per attribute per
// step (two concatenated copies of the input, a 64-fold tiling of the parameters, one bmm + add +
// tanh chain per layer, sigmoid/abs/max/log2 and two permutes) become ONE pass that reads x and
// writes bits (8 B per element); the backward reads x and v_bits and writes v_x (12 B per element)
// and re-evaluates the two tiny MLPs instead of storing their activations.
//
// Per element x[n, c]:  lower/upper = f_p(x -/+ Q_c/2),  f_p: 1 -> W -> ... -> W -> 1 (L hidden
// layers), h <- softplus(M) h + b, h <- h + tanh(F) tanh(h) on all but the last layer;
// sign = -sign(lower + upper); likelihood = |sigmoid(sign upper) - sigmoid(sign lower)| clamped
// from below at `bound` (LowerBound: gradient passes if likelihood >= bound or the incoming
// gradient is negative); bits = -log2(likelihood).
// Parameter set of an element -- the reference's "times = 32" reshape quirk, reproduced because it
// defines the numbers a drop-in must match (oracle/entropy_oracle.py explains it):
//     p(n, c) = (32 c + n / chunk) % C,   chunk = (N + 32 - N % 32) / 32.
//
// Mapping: one workgroup = one (chunk j, channel c) pair and a run of rows, so its parameter set is
// uniform and lives in SGPRs (see entropy_kernel).  In the backward every thread keeps the P
// parameter-gradient sums in registers over its whole loop; DPP wave sums, one LDS meeting point per
// workgroup, P global atomics per workgroup.  At N = 1M, C = 3 the forward moves 24 MB: the kernels
// are bound by the ~150 / ~300 VALU + transcendental instructions per element, not by HBM.
#include "gs_common.h"

#include <cmath>

namespace {

constexpr int ENT_MAX_C = 32;    // channels per call (the reference default is 32; it uses 1, 3, 4)
constexpr int ENT_ROWS = 4096;   // rows of x per block (upper bound; shrinks for small inputs)
constexpr int ENT_TIMES = 32;    // the reference's reshape factor

template <int L, int W>
struct EntLayout {
    // per parameter set: [A0 (W) | b0 (W) | f0 (W)] [A_i (W*W) | b_i (W) | f_i (W)] x (L-1) [A_L (W) | b_L (1)]
    static constexpr int P = 3 * W + (L - 1) * (W * W + 2 * W) + W + 1;
    __host__ __device__ static constexpr int off_hidden(int i) { return i == 0 ? 0 : 3 * W + (i - 1) * (W * W + 2 * W); }
    static constexpr int off_last = 3 * W + (L - 1) * (W * W + 2 * W);
};

GS_DEV float softplus_t(float x) { return x > 20.f ? x : log1pf(expf(x)); } // torch: beta 1, threshold 20 (once per block: libm)
// Hot-loop transcendentals on the hardware units (v_exp_f32 / v_rcp_f32 / v_log_f32, ~1 ulp): the
// kernel evaluates 12-18 tanh per element, libm's tanhf made it transcendental-bound (measured 6% of HBM).
GS_DEV float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
GS_DEV float sigmoid_t(float x) { return __builtin_amdgcn_rcpf(1.f + exp_fast(-x)); }
GS_DEV float tanh_fast(float x) { // 1 - 2 / (1 + e^{2x}); saturates correctly at +-inf, abs error ~1e-7
    return 1.f - 2.f * __builtin_amdgcn_rcpf(1.f + exp_fast(2.f * x));
}

// transformed parameter (softplus / identity / tanh by position) and its derivative w.r.t. the raw one
template <int L, int W>
GS_DEV void transform_param(int k, float raw, float &val, float &deriv) {
    using LY = EntLayout<L, W>;
    int kind; // 0 matrix, 1 bias, 2 factor
    if (k >= LY::off_last) {
        kind = (k - LY::off_last) < W ? 0 : 1;
    } else if (k < 3 * W) {
        kind = k / W;
    } else {
        const int r = (k - 3 * W) % (W * W + 2 * W);
        kind = r < W * W ? 0 : (r < W * W + W ? 1 : 2);
    }
    if (kind == 0) {
        val = softplus_t(raw);
        deriv = raw > 20.f ? 1.f : sigmoid_t(raw);
    } else if (kind == 1) {
        val = raw;
        deriv = 1.f;
    } else {
        val = tanhf(raw);
        deriv = 1.f - val * val;
    }
}

// logits = f_p(x); hs[i] = input of hidden layer i+1 (output of hidden layer i), ts[i] = tanh of
// hidden layer i's pre-activation.  `par` points at the transformed parameter set in LDS.
template <int L, int W>
GS_DEV float mlp_eval(float x, const float *par, float (&hs)[L][W], float (&ts)[L][W]) {
    using LY = EntLayout<L, W>;
#pragma unroll
    for (int o = 0; o < W; ++o) {
        const float z = par[o] * x + par[W + o];
        const float t = tanh_fast(z);
        ts[0][o] = t;
        hs[0][o] = z + par[2 * W + o] * t;
    }
#pragma unroll
    for (int i = 1; i < L; ++i) {
        const float *q = par + LY::off_hidden(i);
#pragma unroll
        for (int o = 0; o < W; ++o) {
            float z = q[W * W + o];
#pragma unroll
            for (int k = 0; k < W; ++k) z += q[o * W + k] * hs[i - 1][k];
            const float t = tanh_fast(z);
            ts[i][o] = t;
            hs[i][o] = z + q[W * W + W + o] * t;
        }
    }
    const float *q = par + LY::off_last;
    float out = q[W];
#pragma unroll
    for (int k = 0; k < W; ++k) out += q[k] * hs[L - 1][k];
    return out;
}

// back-propagate g (= d loss / d logits) through one evaluation; adds to the P register sums, returns d/dx
template <int L, int W>
GS_DEV float mlp_grad(float x, float g_out, const float *par, const float (&hs)[L][W], const float (&ts)[L][W],
                      float (&acc)[EntLayout<L, W>::P]) {
    using LY = EntLayout<L, W>;
    float g[W];
    {
        const float *q = par + LY::off_last;
        acc[LY::off_last + W] += g_out; // b_L
#pragma unroll
        for (int k = 0; k < W; ++k) {
            acc[LY::off_last + k] += g_out * hs[L - 1][k];
            g[k] = q[k] * g_out;
        }
    }
#pragma unroll
    for (int i = L - 1; i >= 1; --i) {
        const float *q = par + LY::off_hidden(i);
        const int base = LY::off_hidden(i);
        float gp[W];
#pragma unroll
        for (int k = 0; k < W; ++k) gp[k] = 0.f;
#pragma unroll
        for (int o = 0; o < W; ++o) {
            const float t = ts[i][o], f = q[W * W + W + o];
            acc[base + W * W + W + o] += g[o] * t;            // tanh(F)
            const float gz = g[o] * (1.f + f * (1.f - t * t));
            acc[base + W * W + o] += gz;                      // bias
#pragma unroll
            for (int k = 0; k < W; ++k) {
                acc[base + o * W + k] += gz * hs[i - 1][k];   // matrix
                gp[k] += q[o * W + k] * gz;
            }
        }
#pragma unroll
        for (int k = 0; k < W; ++k) g[k] = gp[k];
    }
    float gx = 0.f;
#pragma unroll
    for (int o = 0; o < W; ++o) {
        const float t = ts[0][o], f = par[2 * W + o];
        acc[2 * W + o] += g[o] * t;
        const float gz = g[o] * (1.f + f * (1.f - t * t));
        acc[W + o] += gz;
        acc[o] += gz * x;
        gx += par[o] * gz;
    }
    return gx;
}

struct EntArgs {
    uint64_t n;
    uint32_t channels, chunk, rows_per_block, replicas;
    const float *x, *half_q, *params;
    float bound;
};

// likelihood pieces shared by forward and backward
struct Lik {
    float s, su, sl, lik;
};
GS_DEV Lik likelihood(float lower, float upper) {
    Lik r;
    const float sum = lower + upper;
    r.s = sum > 0.f ? -1.f : (sum < 0.f ? 1.f : 0.f); // -sign(lower + upper), sign(0) = 0
    r.su = sigmoid_t(r.s * upper);
    r.sl = sigmoid_t(r.s * lower);
    r.lik = fabsf(r.su - r.sl);
    return r;
}

// 64-lane sum with DPP row shifts + row broadcasts; the total lands in lane 63
GS_DEV float wave_sum63(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x111, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x112, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x114, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x118, 0xf, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x142, 0xa, 0xf, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x143, 0xc, 0xf, false));
    return v;
}

// grid: x = run of rows inside a chunk, y = chunk j (0..31), z = channel c.  One (chunk, channel) pair
// has ONE parameter set, so inside a workgroup the parameters are wave-uniform: they are transformed
// once (softplus / tanh), parked in LDS and pulled into SGPRs -- the MLP's multiply-adds take them as
// scalar operands and no vector register holds a parameter (the first version, with the channel varying
// across lanes, needed 166-234 VGPRs and ran at 2-3 waves per SIMD).  A wave reads x[row, c] for 64
// consecutive rows (stride C floats: the C workgroups of a row run share the lines through L2).
template <int L, int W, bool BWD>
__global__ void __launch_bounds__(256) entropy_kernel(EntArgs a, float *__restrict__ bits, const float *__restrict__ v_bits,
                                                       float *__restrict__ v_x, float *__restrict__ v_params) {
    using LY = EntLayout<L, W>;
    constexpr int P = LY::P;
    __shared__ float s_par[P];
    __shared__ float s_der[BWD ? P : 1];
    __shared__ float s_sum[BWD ? P : 1];
    const uint32_t C = a.channels;
    const uint32_t tid = threadIdx.x;
    const uint32_t j = blockIdx.y, c = blockIdx.z;
    const uint32_t p = (ENT_TIMES * c + j) % C;
    for (uint32_t k = tid; k < (uint32_t)P; k += blockDim.x) {
        float val, der;
        transform_param<L, W>((int)k, a.params[p * P + k], val, der);
        s_par[k] = val;
        if (BWD) {
            s_der[k] = der;
            s_sum[k] = 0.f;
        }
    }
    __syncthreads();
    float par[P]; // wave-uniform: lives in SGPRs
#pragma unroll
    for (int k = 0; k < P; ++k) par[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_par[k])));
    const uint64_t row0 = (uint64_t)j * a.chunk + (uint64_t)blockIdx.x * a.rows_per_block;
    uint64_t row_end = row0 + a.rows_per_block;
    if (row_end > (uint64_t)(j + 1) * a.chunk) row_end = (uint64_t)(j + 1) * a.chunk;
    if (row_end > a.n) row_end = a.n;
    const float hq = a.half_q[c];
    float acc[BWD ? P : 1];
    if (BWD) {
#pragma unroll
        for (int k = 0; k < P; ++k) acc[k] = 0.f;
    }
    for (uint64_t row = row0 + tid; row < row_end; row += blockDim.x) {
        const uint64_t e = row * C + c;
        const float x = a.x[e];
        float hs_l[L][W], ts_l[L][W], hs_u[L][W], ts_u[L][W];
        const float xl = x - hq, xu = x + hq;
        const float lower = mlp_eval<L, W>(xl, par, hs_l, ts_l);
        const float upper = mlp_eval<L, W>(xu, par, hs_u, ts_u);
        const Lik lk = likelihood(lower, upper);
        const float lik_b = fmaxf(lk.lik, a.bound);
        if constexpr (!BWD) {
            bits[e] = -__builtin_amdgcn_logf(lik_b); // v_log_f32 = log2
        } else {
            const float vb = v_bits[e];
            const float g_lik_b = -vb * __builtin_amdgcn_rcpf(0.6931471805599453f * lik_b);
            const bool pass = (lk.lik >= a.bound) || (g_lik_b < 0.f);
            const float d = lk.su - lk.sl;
            const float sg = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
            const float g_diff = pass ? g_lik_b * sg : 0.f;
            const float g_upper = g_diff * lk.s * lk.su * (1.f - lk.su);
            const float g_lower = -g_diff * lk.s * lk.sl * (1.f - lk.sl);
            float gx = mlp_grad<L, W>(xl, g_lower, par, hs_l, ts_l, acc);
            gx += mlp_grad<L, W>(xu, g_upper, par, hs_u, ts_u, acc);
            v_x[e] = gx;
        }
    }
    if constexpr (BWD) {
        // every lane of the workgroup owns the same parameter set: plain wave sums, 4 waves meet in LDS,
        // one global atomic per value and workgroup into one of `replicas` copies (same-address device
        // atomics serialise at the memory side)
#pragma unroll
        for (int k = 0; k < P; ++k) {
            const float tot = wave_sum63(acc[k]);
            if ((tid & 63u) == 63u) atomicAdd(&s_sum[k], tot);
        }
        __syncthreads();
        const size_t rep = (size_t)((blockIdx.x + (blockIdx.y + blockIdx.z * gridDim.y) * gridDim.x) % a.replicas) * (C * P);
        for (uint32_t k = tid; k < (uint32_t)P; k += blockDim.x) {
            const float v = s_sum[k] * s_der[k];
            if (v != 0.f) unsafeAtomicAdd(v_params + rep + p * P + k, v);
        }
    }
}

template <int L, int W>
int32_t launch_entropy(const EntArgs &a_in, bool bwd, float *bits, const float *v_bits, float *v_x, float *v_params, hipStream_t st) {
    EntArgs a = a_in;
    // ~1000+ workgroups (4 per CU) but runs long enough that the per-workgroup parameter transform and
    // (backward) reduction stay small
    uint32_t rows = ENT_ROWS;
    while (rows > 256 && (uint64_t)ENT_TIMES * a.channels * gs_div_up(a.chunk, rows) < 1024) rows /= 2;
    a.rows_per_block = rows;
    dim3 grid(gs_div_up(a.chunk, rows), ENT_TIMES, a.channels);
    if (bwd)
        hipLaunchKernelGGL((entropy_kernel<L, W, true>), grid, dim3(256), 0, st, a, bits, v_bits, v_x, v_params);
    else
        hipLaunchKernelGGL((entropy_kernel<L, W, false>), grid, dim3(256), 0, st, a, bits, v_bits, v_x, v_params);
    return 0;
}

int32_t dispatch_entropy(uint32_t L, uint32_t W, const EntArgs &a, bool bwd, float *bits, const float *v_bits, float *v_x,
                         float *v_params, hipStream_t st) {
#define ENT_CASE(l, w) \
    if (L == l && W == w) return launch_entropy<l, w>(a, bwd, bits, v_bits, v_x, v_params, st);
    ENT_CASE(1, 1) ENT_CASE(1, 2) ENT_CASE(1, 3) ENT_CASE(1, 4)
    ENT_CASE(2, 1) ENT_CASE(2, 2) ENT_CASE(2, 3) ENT_CASE(2, 4)
    ENT_CASE(3, 1) ENT_CASE(3, 2) ENT_CASE(3, 3) ENT_CASE(3, 4)
    ENT_CASE(4, 1) ENT_CASE(4, 2) ENT_CASE(4, 3) ENT_CASE(4, 4)
#undef ENT_CASE
    gs_set_error("gs_entropy_factorized: unsupported filters (hidden layers %u, width %u): 1..4 layers of equal width 1..4", L, W);
    return 1;
}

uint32_t params_per_channel(uint32_t L, uint32_t W) { return 3 * W + (L - 1) * (W * W + 2 * W) + W + 1; }

} // namespace

extern "C" uint32_t gs_entropy_factorized_params_per_channel(uint32_t hidden_layers, uint32_t hidden_width) {
    if (hidden_layers < 1 || hidden_layers > 4 || hidden_width < 1 || hidden_width > 4) return 0;
    return params_per_channel(hidden_layers, hidden_width);
}

extern "C" int32_t gs_entropy_factorized_fwd(uint64_t n, uint32_t channels, uint32_t hidden_layers, uint32_t hidden_width,
                                             const float *x, const float *half_q, const float *params,
                                             float likelihood_bound, float *bits, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(x && half_q && params && bits, "null pointer");
    GS_CHECK_ARG(channels >= 1 && channels <= (uint32_t)ENT_MAX_C, "channels must be in 1..32");
    EntArgs a;
    a.n = n;
    a.channels = channels;
    a.chunk = (uint32_t)((n + (ENT_TIMES - n % ENT_TIMES)) / ENT_TIMES);
    a.rows_per_block = 0;
    a.x = x; a.half_q = half_q; a.params = params;
    a.bound = likelihood_bound;
    a.replicas = 1;
    int32_t rc = dispatch_entropy(hidden_layers, hidden_width, a, false, bits, nullptr, nullptr, nullptr, (hipStream_t)stream);
    if (rc != 0) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}

extern "C" int32_t gs_entropy_factorized_bwd(uint64_t n, uint32_t channels, uint32_t hidden_layers, uint32_t hidden_width,
                                             const float *x, const float *half_q, const float *params,
                                             float likelihood_bound, const float *v_bits, float *v_x, float *v_params,
                                             uint32_t replicas, gs_stream_t stream) {
    if (n == 0) return 0;
    GS_CHECK_ARG(x && half_q && params && v_bits && v_x && v_params, "null pointer");
    GS_CHECK_ARG(replicas >= 1, "replicas must be >= 1");
    GS_CHECK_ARG(channels >= 1 && channels <= (uint32_t)ENT_MAX_C, "channels must be in 1..32");
    EntArgs a;
    a.n = n;
    a.channels = channels;
    a.chunk = (uint32_t)((n + (ENT_TIMES - n % ENT_TIMES)) / ENT_TIMES);
    a.rows_per_block = 0;
    a.x = x; a.half_q = half_q; a.params = params;
    a.bound = likelihood_bound;
    a.replicas = replicas;
    int32_t rc = dispatch_entropy(hidden_layers, hidden_width, a, true, nullptr, v_bits, v_x, v_params, (hipStream_t)stream);
    if (rc != 0) return rc;
    GS_CHECK_LAUNCH();
    return 0;
}
