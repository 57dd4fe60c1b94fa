// A/B for "the compositing backward's cross-lane sums on the matrix pipe" (round-3 verdict, item 4).
//
// The shipped raster_seg_bwd_kernel spends ~47 VALU issue units per record (splat) on the 64-lane reduction of its nine
// per-splat sums (6 permlane swaps + 14 DPP adds + LDS hand-over) and ~11 per (quadrant, record) pass on accumulating them.
// The proposal: stage v_sigma / fac for 16 records through LDS and contract them with pixel-only bases on MFMA --
//     moments[16 records x 6] = v_sigma[16 x 256 pixels] . Phi[256 x 6],   colour sums[16 x 3] = fac[16 x 256] . v_out[256 x 3]
// (256 = 64 lanes x 4 quadrants per wave).  With v_mfma_f32_16x16x4_f32 that is 2 x 64 = 128 MFMAs per 16 records = 8 per record;
// with bf16 operands split into hi + lo parts (Phi is exact in bf16, v_out is not): (2 + 3) x 16 v_mfma_f32_16x16x16_bf16 = 5 per
// record, or 2.5 v_mfma_f32_16x16x32_bf16.
//
// This program measures what those MFMAs cost NEXT TO a saturated VALU at the shipped kernel's occupancy (one wave per
// workgroup, 5 per SIMD): (a) VALU work alone, (b) the same VALU work with k MFMAs interleaved per 64 VALU instructions, (c) the
// MFMAs alone, and (d) the LDS hand-over a transposed A operand needs (one ds_write_b64 per pass, one ds_read_b64 per MFMA).
// marginal cost of an MFMA = (t_b - t_a) / number of MFMAs, to be compared with the 2-cycle issue unit of a plain v_fma.
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_reduce_ab tools/mfma_reduce_ab.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef float f4 __attribute__((ext_vector_type(4)));
typedef short s4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

#define REP 2048

// MODE bits: 1 = VALU block (64 independent-ish v_fma), 2 = f32 MFMAs, 4 = bf16 16x16x16, 8 = bf16 16x16x32, 16 = LDS hand-over
template <int MODE, int NM>
__global__ void __launch_bounds__(64, 5) k(float *out, float seed) {
    __shared__ float s_buf[64 * 4];
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x + i;
    f4 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[i] = (f4){0.f, 0.f, 0.f, 0.f};
    const float m = 0.999f, c = 1e-3f;
    float av = seed, bv = 1.f + seed;
    s4 a16 = {1, 2, 3, 4}, b16 = {5, 6, 7, 8};
    bf8 a32, b32;
#pragma unroll
    for (int i = 0; i < 8; ++i) { a32[i] = (__bf16)(float)(i + 1); b32[i] = (__bf16)(float)(i + 2); }
    for (int r = 0; r < REP; ++r) {
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) { // 8 blocks of 8 fma = 64 VALU per iteration, MFMAs spread between them
            if (MODE & 1)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]) : "v"(m), "v"(c));
            if (blk < NM) {
                if (MODE & 16) { // transposed hand-over: write my value pair, read the operand pair back
                    reinterpret_cast<float2 *>(s_buf)[threadIdx.x] = make_float2(a[0], a[1]);
                    __builtin_amdgcn_wave_barrier();
                    const float2 t = reinterpret_cast<float2 *>(s_buf)[(threadIdx.x * 17 + blk) & 63];
                    av = t.x; bv = t.y;
                }
                if (MODE & 2) acc[blk & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc[blk & 3], 0, 0, 0);
                if (MODE & 4) acc[blk & 3] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a16, b16, acc[blk & 3], 0, 0, 0);
                if (MODE & 8) acc[blk & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a32, b32, acc[blk & 3], 0, 0, 0);
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += acc[i].x + acc[i].y + acc[i].z + acc[i].w;
    out[blockIdx.x * 64 + threadIdx.x] = s + av + bv;
}

template <int MODE, int NM>
double run(const char *name) {
    const int blocks = 256 * 4 * 5; // 5 waves per SIMD, every SIMD of the chip
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * 64);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NM>), dim3(blocks), dim3(64), 0, 0, out, 1.f);
    hipDeviceSynchronize();
    float best = 1e30f;
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NM>), dim3(blocks), dim3(64), 0, 0, out, 1.f);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    // cycles per iteration and SIMD at 2.4 GHz: 5 waves share a SIMD
    const double cyc = best * 1e-3 * 2.4e9 / (5.0 * REP);
    printf("%-58s %.3f ms  %7.1f cycles per (wave, iteration)\n", name, best, cyc);
    hipFree(out);
    return cyc;
}

int main() {
    const double valu = run<1, 0>("64 v_fma");
    printf("  -> %.2f cycles per v_fma (the 'issue unit' of DESIGN.md is 2)\n", valu / 64);
    const double f32_4 = run<1 | 2, 4>("64 v_fma + 4 v_mfma_f32_16x16x4_f32");
    const double f32_8 = run<1 | 2, 8>("64 v_fma + 8 v_mfma_f32_16x16x4_f32");
    const double f32_only = run<2, 8>("8 v_mfma_f32_16x16x4_f32 alone");
    const double b16_4 = run<1 | 4, 4>("64 v_fma + 4 v_mfma_f32_16x16x16_bf16");
    const double b16_8 = run<1 | 4, 8>("64 v_fma + 8 v_mfma_f32_16x16x16_bf16");
    const double b16_only = run<4, 8>("8 v_mfma_f32_16x16x16_bf16 alone");
    const double b32_4 = run<1 | 8, 4>("64 v_fma + 4 v_mfma_f32_16x16x32_bf16");
    const double b32_only = run<8, 8>("8 v_mfma_f32_16x16x32_bf16 alone");
    const double lds_4 = run<1 | 16, 4>("64 v_fma + 4 (ds_write_b64 + ds_read_b64)");
    const double lds_mf = run<1 | 4 | 16, 4>("64 v_fma + 4 (ds_write_b64 + ds_read_b64 + mfma bf16 x16)");
    printf("\nmarginal cycles per MFMA next to a busy VALU (5 waves per SIMD):\n");
    printf("  f32 16x16x4 : %.1f (4 per iteration) %.1f (8 per iteration); alone %.1f\n", (f32_4 - valu) / 4, (f32_8 - valu) / 8, f32_only / 8);
    printf("  bf16 16x16x16: %.1f / %.1f; alone %.1f\n", (b16_4 - valu) / 4, (b16_8 - valu) / 8, b16_only / 8);
    printf("  bf16 16x16x32: %.1f; alone %.1f\n", (b32_4 - valu) / 4, b32_only / 8);
    printf("  LDS hand-over (write + read b64): %.1f per pair; with the MFMA: %.1f\n", (lds_4 - valu) / 4, (lds_mf - valu) / 4);
    printf("\nper RECORD (the shipped reduction: ~47 issue units = ~%.0f cycles):\n", 47 * valu / 64);
    printf("  f32 plan, 8 MFMA per record         : %.0f cycles\n", 8 * (f32_8 - valu) / 8);
    printf("  bf16 hi+lo plan, 5 MFMA x16 per record: %.0f cycles (+ operand hand-over %.0f)\n", 5 * (b16_8 - valu) / 8, 5 * (lds_4 - valu) / 4);
    printf("  bf16 hi+lo plan, 2.5 MFMA x32 per record: %.0f cycles\n", 2.5 * (b32_4 - valu) / 4);
    return 0;
}
