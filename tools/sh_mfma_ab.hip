// A/B of the SH colour contraction: VALU (1 lane per splat, 48 FMAs) vs MFMA (v_mfma_f32_4x4x1_16b_f32: 16 splats per wave,
// 4 lanes per splat, one outer-product accumulate per basis function).  Both read the same [N,16,3] coefficients and [N,3]
// directions and write [N,3] colours; the basis functions are a cheap stand-in polynomial (identical in both kernels) so that
// the comparison isolates the contraction.  Build + run:  hipcc --offload-arch=gfx950 -O3 tools/sh_mfma_ab.hip -o /tmp/sh_ab && /tmp/sh_ab
// north_star: "MFMA only for the SH-colour evaluation contraction ... choices evidenced by rocprof / measurements".
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

constexpr int K = 16;
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void basis(float x, float y, float z, float *Y) {
    // 16 polynomial "basis" values (degree <= 3), the same ~40 flops in both kernels
    const float xx = x * x, yy = y * y, zz = z * z, xy = x * y, yz = y * z, xz = x * z;
    Y[0] = 0.2820948f; Y[1] = -0.4886025f * y; Y[2] = 0.4886025f * z; Y[3] = -0.4886025f * x;
    Y[4] = 1.0925484f * xy; Y[5] = -1.0925484f * yz; Y[6] = 0.9461747f * zz - 0.3153916f; Y[7] = -1.0925484f * xz;
    Y[8] = 0.5462742f * (xx - yy); Y[9] = -0.5900436f * y * (3.f * xx - yy); Y[10] = 2.8906114f * xy * z;
    Y[11] = 0.4570458f * y * (1.f - 5.f * zz); Y[12] = 0.3731763f * z * (5.f * zz - 3.f); Y[13] = 0.4570458f * x * (1.f - 5.f * zz);
    Y[14] = 1.4453057f * z * (xx - yy); Y[15] = -0.5900436f * x * (xx - 3.f * yy);
}

__global__ void __launch_bounds__(256) sh_valu(int n, const float *__restrict__ dirs, const float *__restrict__ coef, float *__restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float Y[K];
    basis(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2], Y);
    const f4 *row = reinterpret_cast<const f4 *>(coef + (size_t)i * K * 3); // 192 B = 12 x 16 B
    float c[48];
#pragma unroll
    for (int q = 0; q < 12; ++q) { const f4 v = row[q]; c[4 * q] = v.x; c[4 * q + 1] = v.y; c[4 * q + 2] = v.z; c[4 * q + 3] = v.w; }
    float r = 0.f, g = 0.f, b = 0.f;
#pragma unroll
    for (int k = 0; k < K; ++k) { r += Y[k] * c[3 * k]; g += Y[k] * c[3 * k + 1]; b += Y[k] * c[3 * k + 2]; }
    out[3 * i] = r; out[3 * i + 1] = g; out[3 * i + 2] = b;
}

// 4 lanes per splat: lane j of the quad holds channel j of the coefficients (j = 3: zero); block b of the 16 4x4 blocks = splat b
// of the wave.  D_b[i][j] += A_b[i] * B_b[j] with A_b[i] = Y_k for every i, B_b[j] = coef[k][j]: after 16 instructions row 0 of
// the block is the colour.
__global__ void __launch_bounds__(256) sh_mfma(int n, const float *__restrict__ dirs, const float *__restrict__ coef, float *__restrict__ out) {
    const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int s = wave * 16 + (lane >> 2), j = lane & 3;
    const bool ok = s < n;
    float Y[K];
    basis(ok ? dirs[3 * s] : 0.f, ok ? dirs[3 * s + 1] : 0.f, ok ? dirs[3 * s + 2] : 0.f, Y);
    f4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const float bval = (ok && j < 3) ? coef[(size_t)s * K * 3 + 3 * k + j] : 0.f;
        acc = __builtin_amdgcn_mfma_f32_4x4x1f32(Y[k], bval, acc, 0, 0, 0);
    }
    // lane (block b, column j) holds D_b[0..3][j] in acc.x .. acc.w: row 0 = acc.x
    if (ok && j < 3) out[3 * s + j] = acc.x;
}

int main() {
    const int n = 1 << 20;
    std::vector<float> hd(3 * (size_t)n), hc((size_t)n * K * 3);
    for (size_t i = 0; i < hd.size(); ++i) hd[i] = (float)((i * 2654435761u) % 2000) / 1000.f - 1.f;
    for (size_t i = 0; i < hc.size(); ++i) hc[i] = (float)((i * 40503u) % 1000) / 1000.f - 0.5f;
    float *d, *c, *o1, *o2;
    hipMalloc(&d, hd.size() * 4); hipMalloc(&c, hc.size() * 4); hipMalloc(&o1, 3 * (size_t)n * 4); hipMalloc(&o2, 3 * (size_t)n * 4);
    hipMemcpy(d, hd.data(), hd.size() * 4, hipMemcpyHostToDevice); hipMemcpy(c, hc.data(), hc.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double bytes = (double)n * (12 + 192 + 12);
    for (int variant = 0; variant < 2; ++variant) {
        float best = 1e9f;
        for (int it = 0; it < 20; ++it) {
            hipEventRecord(e0);
            if (variant == 0) hipLaunchKernelGGL(sh_valu, dim3((n + 255) / 256), dim3(256), 0, 0, n, d, c, o1);
            else hipLaunchKernelGGL(sh_mfma, dim3((n * 4 + 255) / 256), dim3(256), 0, 0, n, d, c, o2);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (it >= 3 && ms < best) best = ms;
        }
        printf("%s: %.1f us, %.0f GB/s (%.0f B/splat, N = %d)\n", variant ? "MFMA 4x4x1 (4 lanes/splat)" : "VALU (1 lane/splat)      ", best * 1e3, bytes / best / 1e6, bytes / n, n);
    }
    std::vector<float> a(3 * (size_t)n), b(3 * (size_t)n);
    hipMemcpy(a.data(), o1, a.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), o2, b.size() * 4, hipMemcpyDeviceToHost);
    double md = 0; for (size_t i = 0; i < a.size(); ++i) md = fmax(md, fabs((double)a[i] - b[i]));
    printf("max |VALU - MFMA| = %.3g\n", md);
    return 0;
}
