// Issue-rate micro-benchmarks for the instruction mix of the compositing kernels (gfx950).
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench tools/ubench.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float float2v __attribute__((ext_vector_type(2)));

#define REP 4096

template <int MODE> __global__ void __launch_bounds__(256) k(float *out, float seed) {
    float a0 = seed + threadIdx.x, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 0.999f, c = 1e-3f;
    for (int r = 0; r < REP; ++r) {
        if (MODE == 0) { // 8 independent v_fma_f32
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else if (MODE == 1) { // 4 independent v_pk_fma_f32 (8 fmas)
            float2v p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, mm = {m, m}, cc = {c, c};
            asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         "v_pk_fma_f32 %0, %0, %4, %5\n v_pk_fma_f32 %1, %1, %4, %5\n v_pk_fma_f32 %2, %2, %4, %5\n v_pk_fma_f32 %3, %3, %4, %5\n"
                         : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc));
            a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y;
        } else if (MODE == 2) { // 8 v_exp_f32
            asm volatile("v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 3) { // 8 v_rcp_f32
            asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 4) { // 8 DPP adds (row_shr:1)
            asm volatile("s_nop 1\n v_add_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %1, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %2, %2, %2 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %3, %3, %3 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %4, %4, %4 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %5, %5, %5 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         "v_add_f32_dpp %6, %6, %6 row_shr:1 row_mask:0xf bank_mask:0xf\n v_add_f32_dpp %7, %7, %7 row_shr:1 row_mask:0xf bank_mask:0xf\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 5) { // 4 permlane32_swap (8 regs)
            asm volatile("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                         "v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane32_swap_b32 %4, %5\n v_permlane32_swap_b32 %6, %7\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
        } else if (MODE == 6) { // 8 v_cndmask / v_cmp pairs: 4 cmp + 4 cndmask
            asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cmp_gt_f32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n"
                         "v_cmp_gt_f32 vcc, %4, %5\n v_cndmask_b32 %4, %4, %5, vcc\n v_cmp_gt_f32 vcc, %6, %7\n v_cndmask_b32 %6, %6, %7, vcc\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : : "vcc");
        } else if (MODE == 7) { // 8 v_add_f32 (non-fma)
            asm volatile("v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n"
                         "v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(c));
        } else if (MODE == 8) { // dependent chain: 8 v_fma on ONE accumulator
            asm volatile("v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         "v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n v_fma_f32 %0, %0, %1, %2\n"
                         : "+v"(a0) : "v"(m), "v"(c));
        } else if (MODE == 9) { // 8 v_mfma_f32_4x4x1 (16 blocks) independent accumulators? use 2 accumulators of 4 regs
            typedef float f4 __attribute__((ext_vector_type(4)));
            f4 acc0 = {a0, a1, a2, a3}, acc1 = {a4, a5, a6, a7};
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(m, c, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(m, c, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(m, c, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(m, c, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(m, c, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(m, c, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(m, c, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(m, c, acc1, 0, 0, 0);
            a0 = acc0.x; a1 = acc0.y; a2 = acc0.z; a3 = acc0.w; a4 = acc1.x; a5 = acc1.y; a6 = acc1.z; a7 = acc1.w;
        } else if (MODE == 10) { // 8 v_mul_f32 then readlane-free SALU mix: s_ instructions interleaved
            asm volatile("v_mul_f32 %0, %0, %8\n s_nop 0\n v_mul_f32 %1, %1, %8\n s_nop 0\n v_mul_f32 %2, %2, %8\n s_nop 0\n v_mul_f32 %3, %3, %8\n s_nop 0\n"
                         "v_mul_f32 %4, %4, %8\n s_nop 0\n v_mul_f32 %5, %5, %8\n s_nop 0\n v_mul_f32 %6, %6, %8\n s_nop 0\n v_mul_f32 %7, %7, %8\n s_nop 0\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

template <int MODE> void run(const char *name, int waves_per_simd) {
    const int n_cu = 256;
    const int block = 256;                       // 4 waves = 1 per SIMD
    const int blocks = n_cu * waves_per_simd;    // waves_per_simd blocks per CU
    float *out;
    hipMalloc(&out, sizeof(float) * blocks * block);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(block), 0, 0, out, 1.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(block), 0, 0, out, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // instructions per SIMD = waves_per_simd * REP * 8
    double inst = (double)waves_per_simd * REP * 8;
    double ns_per_inst = ms * 1e6 / inst;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.3f ns/inst/SIMD  (= %.2f cyc @2.4GHz)\n", name, waves_per_simd, ms, ns_per_inst, ns_per_inst * 2.4);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32 x8 indep", w);
        run<1>("v_pk_fma_f32 x8 (16 fma)", w);
        run<7>("v_add_f32 x8", w);
        run<2>("v_exp_f32 x8", w);
        run<3>("v_rcp_f32 x8", w);
        run<4>("v_add_f32_dpp row_shr x8", w);
        run<5>("v_permlane32_swap x8", w);
        run<6>("v_cmp+v_cndmask x4", w);
        run<8>("v_fma_f32 dependent x8", w);
        run<9>("v_mfma_f32_4x4x1 x8", w);
        run<10>("v_mul + s_nop x8", w);
    }
    return 0;
}
