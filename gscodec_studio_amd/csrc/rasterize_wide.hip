// rasterize_wide.hip -- R5b for 5..16 colour channels per launch (feature rendering: the 9-channel render of the reference's
// spacetime trainer, examples/simple_trainer_STG.py:531-551; 17..32 channels as two launches over halves): the depth-segmented
// backward of rasterize.hip (one wave64 per (tile, 256-entry segment) work item of the cost-ordered list, four pixels per lane,
// state restored from the forward's checkpoints, scalar B) with the colours in the LDS record.
//
// Replaces gsplat/cuda/csrc/rasterize_to_pixels_bwd.cu:17-277 for CDIM in {8, 9, 16, 17, 32} (the reference pads the channel
// count up to one of its template instances, gsplat/cuda/_wrapper.py:496-541); same per-pixel arithmetic and decisions as the
// 1..4-channel kernel (see the header of rasterize.hip).
//
// What differs from raster_seg_bwd_kernel:
//   * record = R0 (mx, my, a', b')  R1 (c', log2 o, col0, col1)  R2 (col2, col3, a, b)  R3 (c, o, g, -)  R4.. (col4 ...);
//   * the 6 + CDIM (+ 2 with absgrad) per-splat sums go through butterflies of eight (wave_fold_8: two permlane-swap levels)
//     that share ONE interleaved row reduction (wave_rows_reduce_4 / _6), a tail of <= 4 values through plain DPP chains:
//     2.5 instructions per value instead of 6;
//   * gradients: the geometry (mean2d, conic, opacity, absgrad) into the packed 64-byte rows as before -- the projection
//     backward reads them in place --, the colours into their own dense [n_elems, channels] array, both with lane = (splat
//     slot, component) atomics (one request per splat and line);
//   * a launch covers channels [ch_off, ch_off + cnt): every gradient is LINEAR in the image gradient, so two launches over
//     the two halves of 17..32 channels add up exactly (v_render_alphas rides with the first).  absgrad is not linear: with it
//     more than 16 channels stay on the generic one-pass kernel.
#include "gs_common.h"
#include "rasterize_common.h"
#include "dpp_reduce.h"
#include "rasterize_dev.h"

namespace {

// waves per SIMD the register allocation of the <= 9-channel instances is held to (0: the compiler's choice, 3 at 9 channels;
// 4 measured 1.072 -> 1.037 ms per 9-channel step at config 2, round 5)
#ifndef GS_WIDE_BWD_WAVES
#define GS_WIDE_BWD_WAVES 4
#endif
#ifndef GS_WIDE_BWD_WAVES_HI // the 12- / 16-channel instances (the compiler's choice: 161 / 189 VGPRs = 3 / 2 waves)
#define GS_WIDE_BWD_WAVES_HI 3 // (16 channels: 806 -> 723 us; 4 is out of reach: 13 KB of LDS per wave hold the CU to 12 waves)
#endif
template <int CDIM, bool ABS>
__global__ void __launch_bounds__(GS_WAVE, (GS_WIDE_BWD_WAVES > 0 && CDIM <= 9) ? GS_WIDE_BWD_WAVES : GS_WIDE_BWD_WAVES_HI) raster_seg_bwd_wide_kernel(RasterArgs a, RasterGradArgs ga, int use_v_alpha, SegArgs sg,
                                                                      uint32_t ch_off, uint32_t cnt) {
    static_assert(CDIM > 4 && CDIM <= 16, "5..16 channels per launch");
    constexpr int NC4 = (CDIM - 4 + 3) / 4; // float4s of colours 4..
    constexpr int REC = 4 + NC4;
    constexpr int NV = 6 + CDIM + (ABS ? 2 : 0); // per-splat sums
    constexpr int NGF = NV / 8, REM = NV % 8;    // full butterflies, tail
    constexpr bool TAIL_BFLY = REM > 4;          // tail as a zero-padded butterfly (else REM DPP chains)
    constexpr int NG = NGF + (TAIL_BFLY ? 1 : 0);
    static_assert(NG >= 2 && NG <= 3, "two or three butterflies");
    constexpr int ACCF = 8 * NG + ((!TAIL_BFLY && REM > 0) ? 4 : 0); // floats per accumulator slot
    // slot floats: [Sx Sy Sxx Sxy Syy S0 | C0 C1 C2 ... C(CDIM-1) | Ax Ay]: colour k at 6 + k, absgrad at 6 + CDIM
    __shared__ float4 s_rec[GS_WAVE * REC];
    __shared__ __attribute__((aligned(16))) float s_acc[GS_WAVE * ACCF];
    const uint32_t lane = threadIdx.x;
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    uint32_t n_work = 0;
#pragma unroll 8
    for (int c = 0; c < COST_CLASSES; ++c) n_work += sg.class_count[c];
    if (blockIdx.x >= n_work) return;
    uint32_t r = xcd_remap(blockIdx.x, n_work, a.xcd_group);
    int cls = COST_CLASSES - 1;
    for (; cls > 0; --cls) {
        const uint32_t nc = sg.class_count[cls];
        if (r < nc) break;
        r -= nc;
    }
    const uint2 it = sg.items[(size_t)cls * sg.max_items + r];
    const int32_t seg_k = (int32_t)it.y;
    TileGeom tg = tile_geom(a, it.x);
    if (a.masks != nullptr && !a.masks[tg.lin]) return;
    const int32_t tile_end = tg.range_end;
    tg.range_start = max(tg.range_start, seg_k * sg.seg);
    tg.range_end = min(tg.range_end, (seg_k + 1) * sg.seg);
    const bool from_ckpt = tg.range_end < tile_end;

    const int32_t first = tg.range_end - 1;
    auto fetch = [&](int32_t idx, SplatRaw &s, float *col) {
        s.g = 0;
        s.mx = s.my = s.ca = s.cb = s.cc = s.opac = 0.f;
#pragma unroll
        for (int k = 0; k < CDIM; ++k) col[k] = 0.f;
        if (idx >= tg.range_start) fetch_splat_wide<CDIM>(a, a.flatten_ids[idx], s, col, ch_off, cnt);
    };
    SplatRaw nxt;
    float ncol[CDIM];
    fetch(first - (int32_t)lane, nxt, ncol);

    bool inside[4];
    float T[4], Wq[4], vc[4][CDIM];
    const float px0 = (float)(tg.px0 + lx) + 0.5f, py0 = (float)(tg.py0 + ly) + 0.5f;
    int32_t bin_final[4], q_bin_max[4];
    float qx0[4], qx1[4], qy0[4], qy1[4];
    unsigned q_live = 0;
    const float *bg = a.backgrounds ? a.backgrounds + (size_t)tg.cam * a.channels + ch_off : nullptr;
    const uint32_t CH = a.channels;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const uint32_t ox = lx + 8u * (i & 1), oy = ly + 8u * (i >> 1);
        const uint32_t x = tg.px0 + ox, y = tg.py0 + oy;
        inside[i] = ox < a.tile_size && oy < a.tile_size && x < a.image_width && y < a.image_height;
        const size_t pix = inside[i] ? ((size_t)tg.cam * a.image_height + y) * a.image_width + x : 0;
        const float T_final = inside[i] ? 1.f - ga.render_alphas[pix] : 1.f;
        T[i] = T_final;
        float bg_dot = 0.f;
#pragma unroll
        for (int k = 0; k < CDIM; ++k) {
            vc[i][k] = (inside[i] && (uint32_t)k < cnt) ? ga.v_render_colors[(int64_t)pix * ga.s_vrc_pix + (int64_t)(ch_off + k) * ga.s_vrc_ch] : 0.f;
            if (bg != nullptr && (uint32_t)k < cnt) bg_dot += bg[k] * vc[i][k];
        }
        const float v_a = (inside[i] && use_v_alpha) ? ga.v_render_alphas[pix] : 0.f;
        Wq[i] = T_final * (v_a - bg_dot);
        bin_final[i] = inside[i] ? ga.last_ids[pix] : -1;
        if (from_ckpt && inside[i]) {
            const float *cb = sg.ckpt + (size_t)(seg_k + 1) * (CH + 1) * 256 + i * 64 + lane;
            T[i] = cb[0];
            float bsum = 0.f;
#pragma unroll
            for (int k = 0; k < CDIM; ++k) {
                if ((uint32_t)k < cnt) {
                    float fin = sg.render_colors[pix * CH + ch_off + k];
                    if (bg != nullptr) fin -= T_final * bg[k];
                    bsum += vc[i][k] * (fin - cb[(ch_off + k + 1) * 256]);
                }
            }
            Wq[i] -= bsum;
        }
        q_bin_max[i] = __builtin_amdgcn_readfirstlane(wave_max_i32(bin_final[i]));
        if (q_bin_max[i] >= tg.range_start) q_live |= 1u << i;
        const float X0 = (float)(tg.px0 + 8u * (i & 1)) + 0.5f, Y0 = (float)(tg.py0 + 8u * (i >> 1)) + 0.5f;
        auto sgpr = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
        qx0[i] = sgpr(X0);
        qy0[i] = sgpr(Y0);
        qx1[i] = sgpr(fminf(X0 + 7.f, fminf((float)(tg.px0 + a.tile_size) - 0.5f, (float)a.image_width - 0.5f)));
        qy1[i] = sgpr(fminf(Y0 + 7.f, fminf((float)(tg.py0 + a.tile_size) - 0.5f, (float)a.image_height - 0.5f)));
    }
    if (q_live == 0u) return;
    const int32_t total = first - tg.range_start + 1;
    const int32_t num_batches = (total + GS_WAVE - 1) / GS_WAVE;
    // colour atomics: lane = (slot sub, channel k), `per` slots per instruction
    const uint32_t per = 64u / cnt, csub = lane / cnt, ck = lane - csub * cnt;

    for (int32_t b = 0; b < num_batches; ++b) {
        const int32_t batch_end = first - b * GS_WAVE; // slot t holds list index batch_end - t
        unsigned long long qm[4];
        {
            const SplatRaw s = nxt;
            CullSplat cs;
            const int32_t my_idx = batch_end - (int32_t)lane;
            const bool live = (my_idx >= tg.range_start) && cull_prepare(s, cs);
            const int32_t batch_lo = max(tg.range_start, batch_end - (GS_WAVE - 1));
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const unsigned long long in_play = __ballot(bin_final[i] >= batch_lo);
                qm[i] = 0ull;
                if (((q_live >> i) & 1u) && in_play != 0ull) {
                    const LiveRect lr = live_rect(in_play, qx0[i], qy0[i]);
                    const bool touch = live && (my_idx <= q_bin_max[i]) && rect_touch(s, cs, lr.x0, lr.x1, lr.y0, lr.y1);
                    qm[i] = __ballot(touch);
                }
            }
            float4 *rec = &s_rec[lane * REC];
            rec[0] = make_float4(s.mx, s.my, -0.5f * LOG2E * s.ca, -LOG2E * s.cb);
            rec[1] = make_float4(-0.5f * LOG2E * s.cc, __log2f(s.opac), ncol[0], ncol[1]);
            rec[2] = make_float4(ncol[2], ncol[3], s.ca, s.cb);
            rec[3] = make_float4(s.cc, s.opac, __int_as_float(s.g), 0.f);
#pragma unroll
            for (int j = 0; j < NC4; ++j) {
                auto c = [&](int k) { return k < CDIM ? ncol[k < CDIM ? k : 0] : 0.f; };
                rec[4 + j] = make_float4(c(4 + 4 * j), c(5 + 4 * j), c(6 + 4 * j), c(7 + 4 * j));
            }
        }
        if (b + 1 < num_batches) fetch(first - (b + 1) * GS_WAVE - (int32_t)lane, nxt, ncol);
        __builtin_amdgcn_wave_barrier();

        unsigned long long any = qm[0] | qm[1] | qm[2] | qm[3];
        unsigned long long touched = 0ull;
        while (any) {
            const int t = __builtin_ctzll(any);
            const unsigned long long bit = 1ull << t;
            any &= ~bit;
            const float4 *rec = &s_rec[t * REC];
            const float4 r0 = rec[0], r1 = rec[1], r2 = rec[2];
            const float r3x = ABS ? reinterpret_cast<const float *>(rec + 3)[0] : 0.f;
            float col[CDIM];
            col[0] = r1.z; col[1] = r1.w; col[2] = r2.x; col[3] = r2.y;
#pragma unroll
            for (int j = 0; j < NC4; ++j) {
                const float4 v = rec[4 + j];
                if (4 + 4 * j < CDIM) col[4 + 4 * j < CDIM ? 4 + 4 * j : 0] = v.x;
                if (5 + 4 * j < CDIM) col[5 + 4 * j < CDIM ? 5 + 4 * j : 0] = v.y;
                if (6 + 4 * j < CDIM) col[6 + 4 * j < CDIM ? 6 + 4 * j : 0] = v.z;
                if (7 + 4 * j < CDIM) col[7 + 4 * j < CDIM ? 7 + 4 * j : 0] = v.w;
            }
            // the record is the same for every lane: the colours as wave-uniform SCALARS (one v_readfirstlane each) instead of 64 copies
            // in vector registers -- 9..16 VGPRs back, and with them every spill of these instances (224-408 bytes of scratch per lane
            // before, reloaded in this loop; 0 now): 16 channels 1.400 -> 1.375 ms per step, 9 channels 1.004 -> 0.994
#pragma unroll
            for (int k = 0; k < CDIM; ++k) col[k] = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(col[k])));
            const int32_t idx = batch_end - t;
            float S0 = 0.f, Sx = 0.f, Sy = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f, Ax = 0.f, Ay = 0.f;
            float Cs[CDIM];
#pragma unroll
            for (int k = 0; k < CDIM; ++k) Cs[k] = 0.f;
            float av_sum = 0.f;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!(qm[i] & bit)) continue; // wave-uniform (scalar) branch
                const float dx = r0.x - (px0 + 8.f * (float)(i & 1)), dy = r0.y - (py0 + 8.f * (float)(i >> 1));
                const float lo2 = r1.y;
                const float pl = __builtin_fmaf(dx, __builtin_fmaf(r0.w, dy, r0.z * dx), __builtin_fmaf(r1.x * dy, dy, lo2));
                const float araw = __builtin_amdgcn_exp2f(pl); // = o exp(-sigma)
                const float alpha = fminf(0.999f, araw);
                const bool valid = (idx <= bin_final[i]) && !(pl > lo2) && (alpha >= ALPHA_MIN);
                const float av = valid ? alpha : 0.f;
                av_sum += av;
                const float ra = __builtin_amdgcn_rcpf(1.f - av);
                const float Tn = T[i] * ra;
                const float facv = av * Tn;
                float D = 0.f;
#pragma unroll
                for (int k = 0; k < CDIM; ++k) {
                    D += col[k] * vc[i][k];
                    Cs[k] += facv * vc[i][k];
                }
                const float v_alpha = D * Tn + Wq[i] * ra;
                const float v_sigma = (valid && araw <= 0.999f) ? -araw * v_alpha : 0.f;
                Wq[i] -= facv * D;
                T[i] = Tn;
                const float sdx = v_sigma * dx, sdy = v_sigma * dy;
                S0 += v_sigma;
                Sx += sdx;
                Sy += sdy;
                Sxx += sdx * dx;
                Sxy += sdx * dy;
                Syy += sdy * dy;
                if (ABS) {
                    Ax += fabsf(r2.z * sdx + r2.w * sdy);
                    Ay += fabsf(r2.w * sdx + r3x * sdy);
                }
            }
            if (!__any(av_sum > 0.f)) continue;
            // the value list behind the first butterfly: C2 .. C(CDIM-1) (, Ax, Ay), zero-padded
            auto L = [&](int i) -> float {
                if (i + 2 < CDIM) return Cs[i + 2 < CDIM ? i + 2 : 0];
                if (ABS && i + 2 == CDIM) return Ax;
                if (ABS && i + 2 == CDIM + 1) return Ay;
                return 0.f;
            };
            float lo[3], hi[3];
            wave_fold_8(Sx, Syy, Sxx, Cs[0], Sy, S0, Sxy, Cs[1], lo[0], hi[0]);
            // group j >= 1 holds L[8(j-1) .. 8(j-1)+7]; fold arguments in the order that leaves them in list order in the slot
#pragma unroll
            for (int j = 1; j < NG; ++j) {
                const int o = 8 * (j - 1);
                wave_fold_8(L(o), L(o + 4), L(o + 2), L(o + 6), L(o + 1), L(o + 5), L(o + 3), L(o + 7), lo[j], hi[j]);
            }
            if (NG == 2) wave_rows_reduce_4(lo[0], hi[0], lo[1], hi[1]);
            else wave_rows_reduce_6(lo[0], hi[0], lo[1], hi[1], lo[2], hi[2]);
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f; // the tail through plain chains (totals in lane 63)
            if (!TAIL_BFLY && REM > 0) {
                constexpr int o = 8 * (NGF - 1);
                t0 = L(o);
                if (REM > 1) t1 = L(o + 1);
                if (REM > 2) t2 = L(o + 2);
                if (REM > 3) t3 = L(o + 3);
                if (REM == 1) wave_reduce_sum_1(t0);
                else if (REM == 2) wave_reduce_sum_2(t0, t1);
                else if (REM == 3) wave_reduce_sum_3(t0, t1, t2);
                else wave_reduce_sum_4(t0, t1, t2, t3);
            }
            touched |= bit;
            float *acc = &s_acc[t * ACCF];
            if ((lane & 15u) == 15u) {
                const uint32_t row = lane >> 4; // rows 0..3 hold (v0,v4) (v2,v6) (v1,v5) (v3,v7) of every butterfly
#pragma unroll
                for (int j = 0; j < NG; ++j) reinterpret_cast<float2 *>(acc + 8 * j)[row] = make_float2(lo[j], hi[j]);
            }
            if (!TAIL_BFLY && REM > 0 && lane == GS_WAVE - 1) *reinterpret_cast<float4 *>(acc + 8 * NG) = make_float4(t0, t1, t2, t3);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- flush.  Lane = slot: turn the moment sums into the geometry gradients in place
        if ((touched >> lane) & 1ull) {
            float *acc = &s_acc[lane * ACCF];
            const float4 a0 = reinterpret_cast<const float4 *>(acc)[0];
            const float2 a1 = reinterpret_cast<const float2 *>(acc)[2];
            const float4 e2 = s_rec[lane * REC + 2], e3 = s_rec[lane * REC + 3];
            const float e_ca = e2.z, e_cb = e2.w, e_cc = e3.x, e_op = e3.y;
            reinterpret_cast<float4 *>(acc)[0] = make_float4(e_ca * a0.x + e_cb * a0.y, e_cb * a0.x + e_cc * a0.y, 0.5f * a0.z, a0.w);
            reinterpret_cast<float2 *>(acc)[2] = make_float2(0.5f * a1.x, -a1.y / e_op);
        }
        __builtin_amdgcn_wave_barrier();
        if (ga.packed) {
            // geometry into the packed rows: lane = (slot, component), 8 components per slot (0..5; 6, 7 = absgrad -> columns 10, 11)
            const uint32_t sub = lane >> 3, comp = lane & 7u;
            const bool comp_on = comp < 6u || ABS;
            const uint32_t src = comp < 6u ? comp : 6u + (uint32_t)CDIM + (comp - 6u), dst = comp < 6u ? comp : 4u + comp;
#pragma unroll 1
            for (uint32_t grp = 0; grp < 8u; ++grp) {
                if (((touched >> (grp * 8u)) & 0xffull) == 0ull) continue; // wave-uniform
                const uint32_t slot = grp * 8u + sub;
                if (comp_on && ((touched >> slot) & 1ull)) {
                    const uint32_t g = (uint32_t)__float_as_int(s_rec[slot * REC + 3].z);
                    unsafeAtomicAdd(ga.v_means2d + (size_t)g * 16u + dst, s_acc[slot * ACCF + src]);
                }
            }
        } else if ((touched >> lane) & 1ull) {
            const float *acc = &s_acc[lane * ACCF];
            const size_t g = (size_t)__float_as_int(s_rec[lane * REC + 3].z);
            unsafeAtomicAdd(ga.v_means2d + ga.s_xy * g, acc[0]);
            unsafeAtomicAdd(ga.v_means2d + ga.s_xy * g + 1, acc[1]);
            unsafeAtomicAdd(ga.v_conics + ga.s_conic * g, acc[2]);
            unsafeAtomicAdd(ga.v_conics + ga.s_conic * g + 1, acc[3]);
            unsafeAtomicAdd(ga.v_conics + ga.s_conic * g + 2, acc[4]);
            unsafeAtomicAdd(ga.v_opacities + ga.s_opac * g, acc[5]);
            if (ABS) {
                unsafeAtomicAdd(ga.v_means2d_abs + ga.s_abs * g, acc[6 + CDIM]);
                unsafeAtomicAdd(ga.v_means2d_abs + ga.s_abs * g + 1, acc[7 + CDIM]);
            }
        }
        // colours: lane = (slot, channel), `per` slots per instruction -- a splat's cnt floats are one contiguous request
#pragma unroll 1
        for (uint32_t s0 = 0; s0 < 64u; s0 += per) {
            if (((touched >> s0) & ((1ull << per) - 1ull)) == 0ull) continue; // wave-uniform
            const uint32_t slot = s0 + csub;
            if (csub < per && slot < 64u && ((touched >> slot) & 1ull)) {
                const size_t g = (size_t)__float_as_int(s_rec[slot * REC + 3].z);
                unsafeAtomicAdd(ga.v_colors + g * ga.s_color + ch_off + ck, s_acc[slot * ACCF + 6u + ck]);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

template <int CDIM>
void launch(const RasterArgs &a, const RasterGradArgs &ga, uint32_t max_items, int use_va, const SegArgs &sg, uint32_t ch_off, uint32_t cnt,
            hipStream_t st) {
    if (ga.v_means2d_abs != nullptr)
        hipLaunchKernelGGL((raster_seg_bwd_wide_kernel<CDIM, true>), dim3(max_items), dim3(GS_WAVE), 0, st, a, ga, use_va, sg, ch_off, cnt);
    else
        hipLaunchKernelGGL((raster_seg_bwd_wide_kernel<CDIM, false>), dim3(max_items), dim3(GS_WAVE), 0, st, a, ga, use_va, sg, ch_off, cnt);
}

} // namespace

// One launch of the segmented backward over channels [ch_off, ch_off + cnt), 5 <= cnt <= 16 (instances 8, 9, 12, 16: the
// smallest one that holds cnt).  `use_va`: v_render_alphas takes part (the first launch of a chunked backward only).
void raster_seg_bwd_wide(const RasterArgs &a, const RasterGradArgs &ga, uint32_t max_items, int use_va, const void *items,
                         const uint32_t *class_count, const float *ckpt, const float *render_colors, int32_t seg, uint32_t ch_off,
                         uint32_t cnt, hipStream_t st) {
    const SegArgs sg = {(const uint2 *)items, class_count, max_items, ckpt, render_colors, seg};
    if (cnt <= 8) launch<8>(a, ga, max_items, use_va, sg, ch_off, cnt, st);
    else if (cnt == 9) launch<9>(a, ga, max_items, use_va, sg, ch_off, cnt, st);
    else if (cnt <= 12) launch<12>(a, ga, max_items, use_va, sg, ch_off, cnt, st);
    else launch<16>(a, ga, max_items, use_va, sg, ch_off, cnt, st);
}
