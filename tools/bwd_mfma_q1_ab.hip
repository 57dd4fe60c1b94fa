// bwd_mfma_q1_ab.hip -- round-5 verdict item 5: the per-splat sums of the compositing backward on the MATRIX pipe, in the
// one-quadrant-per-wave layout (64 pixels per wave, K = 64), against the shipped permlane + DPP reduction.
//
// Per (record, quadrant) the backward needs nine sums over the wave's 64 pixels:
//     S0 = sum v,  Sx = sum v dx,  Sy,  Sxx = sum v dx^2,  Sxy,  Syy      (v = v_sigma[pixel][record], d = mean - pixel)
//     C_k = sum f vc_k,  k = 0..2                                          (f = fac[pixel][record], vc = image gradient)
// With pixel coordinates relative to the quadrant centre the moments are matrix products with a per-PIXEL operand that is
// exact in bf16 (1, px, py, px^2, px py, py^2 with px, py in {-3.5 .. 3.5}):  Mom[6 x 16 records] = P[6 x 64] . V[64 x 16],
// and the d-based sums follow per record (Sx = mx S0 - Spx, Sxx = mx^2 S0 - 2 mx Spx + Spxx, ...).  The colour sums are
// C[3 x 16] = VC^T[3 x 64] . F[64 x 16] with BOTH operands arbitrary fp32.  v_mfma_f32_16x16x16_bf16 takes bf16 operands, so
// fp32 accuracy needs the operands split: 2-way (hi + lo: 16 mantissa bits) or 3-way (hi + mid + lo: 24 bits).
//
// What is timed (5 waves per SIMD on every SIMD, one wave per workgroup, NREC records per wave, same per-record evaluation in
// every mode -- a stand-in for the kernel's alpha / v_sigma arithmetic, ~25 VALU):
//   mode 0  evaluation only (the baseline the others are measured against)
//   mode 1  the shipped reduction: 8 values through the permlane butterfly + the ninth through a DPP chain, per record
//   mode 2  MFMA, 2-way split:  per record 4 ds_write_b16 (V hi / lo, F hi / lo, transposed through LDS) + split VALU;
//           per 16 records 16 ds_read_b64 + 4 x (2 + 3) = 20 MFMAs + the moment conversion
//   mode 3  MFMA, 3-way split:  6 ds_write_b16 per record; per 16 records 24 ds_read_b64 + 4 x (3 + 6) = 36 MFMAs
// and what is checked: the nine sums of the first 64 records of wave 0 against a float64 summation of the SAME fp32
// per-pixel values (dumped by mode 1), relative to max(|sum|, the sum of |terms| x 1e-3).
//
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -I gscodec_studio_amd/csrc -o build_abl/bwd_mfma_q1_ab tools/bwd_mfma_q1_ab.hip
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "dpp_reduce.h"

typedef short v4s __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int NREC = 2048; // records per wave (multiple of 16)
constexpr int DUMP = 64;   // records of wave 0 whose per-pixel values are dumped

struct Rec { float mx, my, a, b, c, lo2, c0, c1; }; // mean (quadrant-centred), conic x -log2(e)/2 ..., log2(opacity), colours 0 / 1 (colour 2 = c0 + c1)

__device__ __forceinline__ unsigned short bf16_rne(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// the per-(pixel, record) evaluation: what the backward computes before it sums (shape and cost of the real thing)
__device__ __forceinline__ void eval(const Rec &r, float px, float py, float T, float W, float vc0, float vc1, float vc2, float &v, float &f,
                                     float &dx, float &dy) {
    dx = r.mx - px;
    dy = r.my - py;
    const float pl = __builtin_fmaf(dx, __builtin_fmaf(r.b, dy, r.a * dx), __builtin_fmaf(r.c * dy, dy, r.lo2));
    const float araw = __builtin_amdgcn_exp2f(pl);
    const float alpha = fminf(0.999f, araw);
    const bool valid = !(pl > r.lo2) && alpha >= (1.f / 255.f);
    const float av = valid ? alpha : 0.f;
    const float ra = __builtin_amdgcn_rcpf(1.f - av);
    const float Tn = T * ra;
    f = av * Tn;
    const float D = r.c0 * vc0 + r.c1 * vc1 + (r.c0 + r.c1) * vc2;
    const float v_alpha = D * Tn + W * ra;
    v = (valid && araw <= 0.999f) ? -araw * v_alpha : 0.f;
}

template <int MODE>
__global__ void __launch_bounds__(64, 5) reduce_kernel(const Rec *__restrict__ recs, float *__restrict__ out, float *__restrict__ dump) {
    constexpr int S = MODE == 3 ? 3 : 2; // planes per matrix: hi, mid (2-way: the rest), lo
    __shared__ unsigned short s_b[MODE >= 2 ? 2 * S : 1][16][64]; // [V planes | F planes][record][pixel]: 8 KB (2-way) / 12 KB (3-way) per wave
    const unsigned lane = threadIdx.x;
    const float px = (float)(lane & 7u) - 3.5f, py = (float)(lane >> 3) - 3.5f;
    // per-pixel "image gradient" and state (fixed per wave, like vc / T / Wq in the kernel)
    const float vc0 = 0.3f + 0.01f * (float)lane, vc1 = -0.7f + 0.02f * (float)(lane & 15u), vc2 = 0.11f * (float)(lane >> 2) - 0.5f;
    const float T = 0.35f + 0.005f * (float)lane, W = 0.2f - 0.003f * (float)lane;
    const Rec *my = recs + (size_t)(blockIdx.x % 64u) * NREC; // 64 different streams
    float sink = 0.f;

    // MFMA operands that do not change: lane l holds A[i = l % 16][k = 4 (l / 16) + 0..3] of every 16-pixel chunk c
    v4s Pm[4];                 // moments: rows 1, px, py, px^2, px py, py^2 (exact in bf16), rows 6.. = 0
    v4s Ch[4], Cm[4], Cl[4];   // colours: rows vc0, vc1, vc2 split hi / mid / lo (mode 2 uses hi and (mid + lo) rounded into one)
    if (MODE >= 2) {
        const unsigned i = lane & 15u, g = lane >> 4;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            short m[4], h[4], md[4], l[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const unsigned p = 16u * c + 4u * g + t;
                const float x = (float)(p & 7u) - 3.5f, y = (float)(p >> 3) - 3.5f;
                const float mom = i == 0 ? 1.f : i == 1 ? x : i == 2 ? y : i == 3 ? x * x : i == 4 ? x * y : i == 5 ? y * y : 0.f;
                m[t] = (short)bf16_rne(mom);
                // the other lanes' vc values: recomputed from the pixel index (in the kernel: one LDS transpose per work item)
                const float w0 = 0.3f + 0.01f * (float)p, w1 = -0.7f + 0.02f * (float)(p & 15u), w2 = 0.11f * (float)(p >> 2) - 0.5f;
                const float w = i == 0 ? w0 : i == 1 ? w1 : i == 2 ? w2 : 0.f;
                const unsigned short hh = bf16_rne(w);
                const float r1 = w - bf16_f32(hh);
                const unsigned short mm = bf16_rne(r1);
                const float r2 = r1 - bf16_f32(mm);
                h[t] = (short)hh;
                md[t] = (short)(MODE == 2 ? bf16_rne(r1) : mm);
                l[t] = (short)bf16_rne(r2);
            }
            Pm[c] = (v4s){m[0], m[1], m[2], m[3]};
            Ch[c] = (v4s){h[0], h[1], h[2], h[3]};
            Cm[c] = (v4s){md[0], md[1], md[2], md[3]};
            Cl[c] = (v4s){l[0], l[1], l[2], l[3]};
        }
    }

    for (int r0 = 0; r0 < NREC; r0 += 16) {
#pragma unroll 1
        for (int j = 0; j < 16; ++j) {
            const Rec r = my[r0 + j];
            float v, f, dx, dy;
            eval(r, px, py, T, W, vc0, vc1, vc2, v, f, dx, dy);
            if (MODE == 0) {
                sink += v + f;
            } else if (MODE == 1) {
                const float sdx = v * dx, sdy = v * dy;
                float S0 = v, Sx = sdx, Sy = sdy, Sxx = sdx * dx, Sxy = sdx * dy, Syy = sdy * dy, C0 = f * vc0, C1 = f * vc1, C2 = f * vc2;
                if (dump != nullptr && blockIdx.x == 0 && r0 + j < DUMP) {
                    dump[((size_t)(r0 + j) * 64 + lane) * 4 + 0] = v;
                    dump[((size_t)(r0 + j) * 64 + lane) * 4 + 1] = f;
                    dump[((size_t)(r0 + j) * 64 + lane) * 4 + 2] = dx;
                    dump[((size_t)(r0 + j) * 64 + lane) * 4 + 3] = dy;
                }
                float lo, hi;
                wave_reduce_sum_8_butterfly_rows(Sx, Syy, Sxx, C0, Sy, S0, Sxy, C1, lo, hi, C2);
                if ((lane & 15u) == 15u && blockIdx.x == 0 && r0 + j < DUMP) {
                    // rows 0..3 hold (Sx, Sy) (Sxx, Sxy) (Syy, S0) (C0, C1); C2 arrives as four row partials
                    float *o = out + (size_t)(r0 + j) * 16;
                    const unsigned row = lane >> 4;
                    o[2 * row] = lo;
                    o[2 * row + 1] = hi;
                    o[8 + row] = C2;
                }
                sink += lo + hi + C2; // (keeps the reduction alive in every wave)
            } else {
                // split and hand over, transposed: s_b[plane][record j][pixel = lane]
                const unsigned short vh = bf16_rne(v), fh = bf16_rne(f);
                const float v1 = v - bf16_f32(vh), f1 = f - bf16_f32(fh);
                const unsigned short vm = bf16_rne(v1), fm = bf16_rne(f1);
                s_b[0][j][lane] = vh;
                s_b[1][j][lane] = vm;
                s_b[S][j][lane] = fh;
                s_b[S + 1][j][lane] = fm;
                if (MODE == 3) {
                    s_b[2][j][lane] = bf16_rne(v1 - bf16_f32(vm));
                    s_b[S + 2][j][lane] = bf16_rne(f1 - bf16_f32(fm));
                }
            }
        }
        if (MODE >= 2) {
            __builtin_amdgcn_wave_barrier();
            // B operand: lane l holds B[k = 4 (l / 16) + 0..3][n = l % 16] of chunk c = 4 consecutive pixels of record l % 16
            const unsigned n = lane & 15u, g = lane >> 4;
            v4f mom = {0.f, 0.f, 0.f, 0.f}, col = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const unsigned off = 16u * c + 4u * g;
                const v4s Vh = *reinterpret_cast<const v4s *>(&s_b[0][n][off]), Vm = *reinterpret_cast<const v4s *>(&s_b[1][n][off]);
                const v4s Fh = *reinterpret_cast<const v4s *>(&s_b[S][n][off]), Fm = *reinterpret_cast<const v4s *>(&s_b[S + 1][n][off]);
                mom = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Pm[c], Vh, mom, 0, 0, 0);
                mom = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Pm[c], Vm, mom, 0, 0, 0);
                col = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Ch[c], Fh, col, 0, 0, 0);
                col = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Ch[c], Fm, col, 0, 0, 0);
                col = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Cm[c], Fh, col, 0, 0, 0);
                if (MODE == 3) {
                    const v4s Vl = *reinterpret_cast<const v4s *>(&s_b[2][n][off]), Fl = *reinterpret_cast<const v4s *>(&s_b[S + 2][n][off]);
                    mom = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Pm[c], Vl, mom, 0, 0, 0);
                    col = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Ch[c], Fl, col, 0, 0, 0);
                    col = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Cl[c], Fh, col, 0, 0, 0);
                    col = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(Cm[c], Fm, col, 0, 0, 0);
                }
            }
            // D[i = 4 (l / 16) + 0..3][n]: lanes 0..15 hold (S0, Spx, Spy, Spxx) of record n, lanes 16..31 (Spxy, Spyy, -, -);
            // the colour sums sit in lanes 0..15 (C0, C1, C2, -).  Pixel moments -> d-based sums, per record:
            const float Spxy = __shfl(mom.x, (int)n + 16, 64), Spyy = __shfl(mom.y, (int)n + 16, 64);
            if (lane < 16u) {
                const Rec r = my[r0 + lane];
                const float S0 = mom.x, Spx = mom.y, Spy = mom.z, Spxx = mom.w;
                const float Sx = r.mx * S0 - Spx, Sy = r.my * S0 - Spy;
                const float Sxx = __builtin_fmaf(r.mx, __builtin_fmaf(r.mx, S0, -2.f * Spx), Spxx);
                const float Syy = __builtin_fmaf(r.my, __builtin_fmaf(r.my, S0, -2.f * Spy), Spyy);
                const float Sxy = __builtin_fmaf(r.mx, __builtin_fmaf(r.my, S0, -Spy), __builtin_fmaf(-r.my, Spx, Spxy));
                if (blockIdx.x == 0 && r0 + (int)lane < DUMP) {
                    float *o = out + (size_t)(r0 + lane) * 16;
                    o[0] = Sx; o[1] = Sy; o[2] = Sxx; o[3] = Sxy; o[4] = Syy; o[5] = S0; o[6] = col.x; o[7] = col.y;
                    o[8] = col.z; o[9] = o[10] = o[11] = 0.f;
                }
                sink += Sx + Sy + Sxx + Sxy + Syy + col.x + col.y + col.z;
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (sink == 1.2345e30f) out[1 << 20] = sink; // (never true: keeps `sink` alive)
}

template <int MODE>
float run(const Rec *d_recs, float *d_out, float *d_dump, int reps) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int grid = 256 * 4 * 5;
    int occ = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, reduce_kernel<MODE>, 64, 0);
    printf("  [mode %d: %d waves resident per CU]\n", MODE, occ);
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(reduce_kernel<MODE>, dim3(grid), dim3(64), 0, 0, d_recs, d_out, d_dump);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    return best;
}

int main() {
    std::vector<Rec> recs((size_t)64 * NREC);
    srand(7);
    auto u = [] { return (float)rand() / (float)RAND_MAX; };
    for (auto &r : recs) {
        const float s1 = 1.5f + 20.f * u() * u(), s2 = s1 * (0.4f + 1.2f * u()), th = 3.14159f * u();
        const float cth = cosf(th), sth = sinf(th);
        const float A = cth * cth * s1 * s1 + sth * sth * s2 * s2, Bc = cth * sth * (s1 * s1 - s2 * s2), Dd = sth * sth * s1 * s1 + cth * cth * s2 * s2;
        const float det = A * Dd - Bc * Bc;
        const float ca = Dd / det, cb = -Bc / det, cc = A / det; // conic
        const float L2E = 1.4426950408889634f;
        r.mx = (u() - 0.5f) * (8.f + 2.f * s1);
        r.my = (u() - 0.5f) * (8.f + 2.f * s1);
        r.a = -0.5f * L2E * ca;
        r.b = -L2E * cb;
        r.c = -0.5f * L2E * cc;
        r.lo2 = log2f(0.05f + 0.95f * u());
        r.c0 = u();
        r.c1 = u();
    }
    Rec *d_recs;
    float *d_out, *d_dump;
    hipMalloc(&d_recs, recs.size() * sizeof(Rec));
    hipMemcpy(d_recs, recs.data(), recs.size() * sizeof(Rec), hipMemcpyHostToDevice);
    hipMalloc(&d_out, ((1 << 20) + 16) * sizeof(float));
    hipMalloc(&d_dump, (size_t)DUMP * 64 * 4 * sizeof(float));

    // ---- accuracy: float64 sums of the dumped fp32 per-pixel values
    std::vector<float> dump((size_t)DUMP * 64 * 4), o1(DUMP * 16), o2(DUMP * 16), o3(DUMP * 16);
    hipMemset(d_out, 0, DUMP * 16 * sizeof(float));
    hipLaunchKernelGGL(reduce_kernel<1>, dim3(1), dim3(64), 0, 0, d_recs, d_out, d_dump);
    hipMemcpy(dump.data(), d_dump, dump.size() * sizeof(float), hipMemcpyDeviceToHost);
    hipMemcpy(o1.data(), d_out, o1.size() * sizeof(float), hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(reduce_kernel<2>, dim3(1), dim3(64), 0, 0, d_recs, d_out, (float *)nullptr);
    hipMemcpy(o2.data(), d_out, o2.size() * sizeof(float), hipMemcpyDeviceToHost);
    hipLaunchKernelGGL(reduce_kernel<3>, dim3(1), dim3(64), 0, 0, d_recs, d_out, (float *)nullptr);
    hipMemcpy(o3.data(), d_out, o3.size() * sizeof(float), hipMemcpyDeviceToHost);
    const char *names[9] = {"Sx", "Sy", "Sxx", "Sxy", "Syy", "S0", "C0", "C1", "C2"};
    double worst[3][9] = {}, mean[3][9] = {};
    int live = 0;
    for (int r = 0; r < DUMP; ++r) {
        double ref[9] = {}, mag[9] = {};
        for (int p = 0; p < 64; ++p) {
            const float *d = &dump[((size_t)r * 64 + p) * 4];
            const double v = d[0], f = d[1], dx = d[2], dy = d[3];
            const double vc0 = 0.3f + 0.01f * (float)p, vc1 = -0.7f + 0.02f * (float)(p & 15), vc2 = 0.11f * (float)(p >> 2) - 0.5f;
            const double t[9] = {v * dx, v * dy, v * dx * dx, v * dx * dy, v * dy * dy, v, f * vc0, f * vc1, f * vc2};
            for (int k = 0; k < 9; ++k) { ref[k] += t[k]; mag[k] += fabs(t[k]); }
        }
        if (mag[5] == 0.0) continue; // nothing valid in this record
        ++live;
        const float *os[3] = {&o1[r * 16], &o2[r * 16], &o3[r * 16]};
        for (int m = 0; m < 3; ++m) {
            double got[9];
            for (int k = 0; k < 8; ++k) got[k] = os[m][k];
            got[8] = m == 0 ? (double)os[m][8] + os[m][9] + os[m][10] + os[m][11] : (double)os[m][8];
            for (int k = 0; k < 9; ++k) {
                const double e = fabs(got[k] - ref[k]) / fmax(fabs(ref[k]), 1e-3 * mag[k] + 1e-30);
                worst[m][k] = fmax(worst[m][k], e);
                mean[m][k] += e;
            }
        }
    }
    printf("accuracy of the nine sums over 64 pixels, %d live records, relative to max(|sum|, 1e-3 sum|terms|), float64 reference:\n", live);
    printf("  %-4s %-26s %-26s %-26s\n", "", "fp32 butterfly (shipped)", "MFMA bf16 2-way split", "MFMA bf16 3-way split");
    for (int k = 0; k < 9; ++k)
        printf("  %-4s worst %.2e mean %.2e   worst %.2e mean %.2e   worst %.2e mean %.2e\n", names[k], worst[0][k], mean[0][k] / live, worst[1][k],
               mean[1][k] / live, worst[2][k], mean[2][k] / live);

    // ---- time
    const float t0 = run<0>(d_recs, d_out, nullptr, 5), t1 = run<1>(d_recs, d_out, nullptr, 5), t2 = run<2>(d_recs, d_out, nullptr, 5),
                t3 = run<3>(d_recs, d_out, nullptr, 5);
    const double cyc = 2.4e9 * 1e-3 / NREC; // cycles per record per ms of kernel time (every SIMD holds 5 waves that run NREC records each)
    printf("\n5 waves per SIMD on every SIMD, %d records per wave (best of 5):\n", NREC);
    printf("  evaluation only                       %.3f ms\n", t0);
    printf("  + shipped butterfly reduction         %.3f ms  -> %.1f cycles per (record, quadrant) per wave slot, x5 waves = %.1f SIMD cycles\n", t1,
           (t1 - t0) * cyc, (t1 - t0) * cyc / 5.0);
    printf("  + MFMA reduction, 2-way split         %.3f ms  -> %.1f / %.1f\n", t2, (t2 - t0) * cyc, (t2 - t0) * cyc / 5.0);
    printf("  + MFMA reduction, 3-way split         %.3f ms  -> %.1f / %.1f\n", t3, (t3 - t0) * cyc, (t3 - t0) * cyc / 5.0);
    printf("\nper RECORD in a one-quadrant-per-wave backward (1.84 touched quadrants per record at config 2) against the shipped kernel's ONE\n"
           "reduction per record over 4 pixels per lane (same butterfly, %.1f SIMD cycles):\n", (t1 - t0) * cyc / 5.0);
    printf("  shipped (4 quadrants per wave)        %.1f SIMD cycles per record\n", (t1 - t0) * cyc / 5.0);
    printf("  q1 + butterfly                        %.1f\n", 1.84 * (t1 - t0) * cyc / 5.0);
    printf("  q1 + MFMA 2-way                       %.1f\n", 1.84 * (t2 - t0) * cyc / 5.0);
    printf("  q1 + MFMA 3-way                       %.1f\n", 1.84 * (t3 - t0) * cyc / 5.0);
    return 0;
}
