// A/B for the compositing backward: "candidate A" of the round-2 verdict (item 4) against the shipped kernel.
//
//   shipped (raster_seg_bwd_kernel): lanes = the 64 pixels of an 8x8 quadrant, loop over the splats of the list; per splat
//       a 64-lane reduction of 9 sums (permlane butterfly + DPP), LDS hand-over, one atomic group per (tile, segment, splat).
//   candidate A: lanes = 64 splats of a batch, loop over the 64 pixels of the quadrant.  Per pixel a multiplicative DPP
//       scan gives every splat the transmittance in front of it, an additive DPP scan the colour behind it (through the
//       identity  sum_{behind} fac D = v_out . (colour_final - colour_through_me)); every lane accumulates ITS OWN nine sums
//       privately -- no cross-lane reduction, no LDS hand-over, one atomic group per lane and batch.
//
// This program measures candidate A's INNER LOOP ALONE -- no staging, no culling, no atomics, pixel state resident in LDS --
// at the shipped kernel's occupancy (one wave per workgroup, 5 per SIMD), i.e. an upper bound on what a kernel built around
// it could reach, in units of "64 splats x 64 pixels" per microsecond.  The shipped kernel's figure for the same unit is
// (quadrant passes / 64) / kernel time from profiles/ (4.6 M passes in 0.28 ms = ~255 units/us at BASELINE config 2), and
// that figure INCLUDES its staging, culling, reductions and atomics.
// Build: hipcc --offload-arch=gfx950 -O3 -o bwd_scan_ab tools/bwd_scan_ab.hip ; run on the GPU box.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define DEV __device__ __forceinline__


template <int CTRL, int ROW_MASK>
DEV float dppf(float old, float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}

// inclusive scans over the 64 lanes (row_shr 1/2/4/8 inside rows of 16, then row_bcast 15 / 31 across rows)
DEV float scan_add(float v) {
    v += dppf<0x111, 0xf>(0.f, v);
    v += dppf<0x112, 0xf>(0.f, v);
    v += dppf<0x114, 0xf>(0.f, v);
    v += dppf<0x118, 0xf>(0.f, v);
    v += dppf<0x142, 0xa>(0.f, v);
    v += dppf<0x143, 0xc>(0.f, v);
    return v;
}
DEV float scan_mul(float v) {
    v *= dppf<0x111, 0xf>(1.f, v);
    v *= dppf<0x112, 0xf>(1.f, v);
    v *= dppf<0x114, 0xf>(1.f, v);
    v *= dppf<0x118, 0xf>(1.f, v);
    v *= dppf<0x142, 0xa>(1.f, v);
    v *= dppf<0x143, 0xc>(1.f, v);
    return v;
}

struct Rec { // one splat, as the shipped kernel stages it
    float mx, my, a, b, c, lo2, col0, col1, col2;
    int idx;
};

constexpr float ALPHA_MIN = 1.f / 255.f;

__global__ void __launch_bounds__(64, 5) scan_bwd_kernel(const Rec *__restrict__ recs, const float *__restrict__ pixels, int batches,
                                                         float *__restrict__ out) {
    __shared__ float4 s_pix[64 * 2]; // per pixel: (px, py, T, Wprefix) (vc0, vc1, vc2, bin_final)
    const int lane = threadIdx.x;
    const float4 *pp = reinterpret_cast<const float4 *>(pixels) + ((size_t)blockIdx.x % 64) * 128;
    s_pix[lane] = pp[lane];
    s_pix[64 + lane] = pp[64 + lane];
    __builtin_amdgcn_wave_barrier();
    float S0 = 0.f, Sx = 0.f, Sy = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f, C0 = 0.f, C1 = 0.f, C2 = 0.f;
    for (int b = 0; b < batches; ++b) {
        const Rec r = recs[((size_t)blockIdx.x * batches + b) % 4096 * 64 + lane]; // (a real kernel stages through LDS + culls here)
#pragma unroll 2
        for (int p = 0; p < 64; ++p) {
            const float4 q0 = s_pix[2 * p], q1 = s_pix[2 * p + 1]; // wave-uniform addresses: LDS broadcast reads
            const float dx = r.mx - q0.x, dy = r.my - q0.y;
            const float pl = __builtin_fmaf(dx, __builtin_fmaf(r.b, dy, r.a * dx), __builtin_fmaf(r.c * dy, dy, r.lo2));
            const float araw = __builtin_amdgcn_exp2f(pl);
            const float alpha = fminf(0.999f, araw);
            const bool valid = (r.idx <= __float_as_int(q1.w)) && !(pl > r.lo2) && (alpha >= ALPHA_MIN);
            const float av = valid ? alpha : 0.f;
            const float om = 1.f - av;
            const float Pinc = scan_mul(om);                          // prod_{j <= me} (1 - a_j)
            const float Pex = dppf<0x138, 0xf>(1.f, Pinc);            // wave_shr:1 -> exclusive
            const float Ti = q0.z * Pex;                              // transmittance in front of me
            const float fac = av * Ti;
            const float D = r.col0 * q1.x + r.col1 * q1.y + r.col2 * q1.z;
            const float Ginc = scan_add(fac * D);                     // colour (dotted with v_out) through me, this batch
            // colour behind me = (final - prefix carried in q0.w) - Ginc  ->  v_alpha = D T + (Tw - behind) / (1 - a)
            const float ra = __builtin_amdgcn_rcpf(om);
            const float v_alpha = __builtin_fmaf(D, Ti, (q0.w + Ginc) * ra);
            const float v_sigma = (valid && araw <= 0.999f) ? -araw * v_alpha : 0.f;
            const float sdx = v_sigma * dx, sdy = v_sigma * dy;
            S0 += v_sigma;
            Sx += sdx;
            Sy += sdy;
            Sxx = __builtin_fmaf(sdx, dx, Sxx);
            Sxy = __builtin_fmaf(sdx, dy, Sxy);
            Syy = __builtin_fmaf(sdy, dy, Syy);
            C0 = __builtin_fmaf(fac, q1.x, C0);
            C1 = __builtin_fmaf(fac, q1.y, C1);
            C2 = __builtin_fmaf(fac, q1.z, C2);
            // carry the pixel's state to the next batch: T <- T * prod(all), prefix <- prefix + sum(all)
            const float Tn = q0.z * __shfl(Pinc, 63, 64), Wn = q0.w + __shfl(Ginc, 63, 64);
            if (lane == 0) s_pix[2 * p] = make_float4(q0.x, q0.y, Tn, Wn);
        }
        __builtin_amdgcn_wave_barrier();
    }
    out[(size_t)blockIdx.x * 64 + lane] = S0 + Sx + Sy + Sxx + Sxy + Syy + C0 + C1 + C2;
}

int main() {
    const int n_wg = 256 * 4 * 5 * 4, batches = 16; // 4 rounds of full residency
    std::vector<Rec> h_recs(4096 * 64);
    std::vector<float> h_pix(64 * 128 * 4);
    srand(1);
    auto rnd = [] { return (float)rand() / (float)RAND_MAX; };
    for (auto &r : h_recs) {
        r.mx = 8.f * rnd(); r.my = 8.f * rnd();
        const float s = 0.02f + 0.3f * rnd();
        r.a = -0.72f * s; r.b = 0.1f * s; r.c = -0.72f * s; r.lo2 = -3.f * rnd();
        r.col0 = rnd(); r.col1 = rnd(); r.col2 = rnd(); r.idx = rand() % 1000;
    }
    for (size_t i = 0; i < h_pix.size(); i += 8) {
        const int p = (int)((i / 8) % 64);
        h_pix[i] = (float)(p % 8) + 0.5f; h_pix[i + 1] = (float)(p / 8) + 0.5f; h_pix[i + 2] = 1.f; h_pix[i + 3] = 0.f;
        h_pix[i + 4] = rnd(); h_pix[i + 5] = rnd(); h_pix[i + 6] = rnd();
        const int bf = 500 + rand() % 500;
        h_pix[i + 7] = *reinterpret_cast<const float *>(&bf);
    }
    // layout fix: s_pix[2p] and s_pix[2p + 1] are consecutive float4s -> the host array above already interleaves them
    Rec *d_recs; float *d_pix, *d_out;
    hipMalloc(&d_recs, h_recs.size() * sizeof(Rec));
    hipMalloc(&d_pix, h_pix.size() * sizeof(float));
    hipMalloc(&d_out, (size_t)n_wg * 64 * sizeof(float));
    hipMemcpy(d_recs, h_recs.data(), h_recs.size() * sizeof(Rec), hipMemcpyHostToDevice);
    hipMemcpy(d_pix, h_pix.data(), h_pix.size() * sizeof(float), hipMemcpyHostToDevice);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(scan_bwd_kernel, dim3(n_wg), dim3(64), 0, 0, d_recs, d_pix, batches, d_out);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double units = (double)n_wg * batches;
        printf("candidate A inner loop: %d workgroups x %d batches in %.3f ms = %.1f units (64 splats x 64 pixels) per us\n", n_wg, batches, ms,
               units / (ms * 1e3));
    }
    printf("shipped raster_seg_bwd_kernel at BASELINE config 2: 4.6 M quadrant passes / 64 in ~0.28 ms = ~255 units/us, staging, culling,\n"
           "reductions and atomics included (profiles/r03_kernel_stats.csv)\n");
    return 0;
}
