// rasterize.hip -- R5: per-tile front-to-back alpha compositing, fwd + bwd, wave-per-tile
// kernels for gfx950.
//
// Replaces gsplat/cuda/csrc/rasterize_to_pixels_fwd.cu:16-185 and
// rasterize_to_pixels_bwd.cu:17-277.  Same per-pixel arithmetic and decisions
// (alpha = min(0.999, o * exp(-sigma)); skip sigma < 0 or alpha < 1/255; exclusive stop when
// T (1 - alpha) <= 1e-4; last_ids; alpha-clamp gradient gate), different mapping:
//
//   * ONE WAVE64 OWNS ONE 16x16 TILE; lane (lx, ly) of an 8x8 grid owns FOUR pixels, one in
//     each 8x8 quadrant.  No workgroup barriers (the reference's 256-thread block needs two
//     __syncthreads per 256 splats); a splat record is fetched from LDS once per 256 pixel
//     evaluations (register blocking over pixels: with one pixel per lane the broadcast
//     ds_read_b128 traffic alone would saturate the LDS pipe); the four pixels give four
//     independent dependency chains per lane, and in the backward the cross-lane reduction
//     is paid once per (tile, splat) on partial sums of 4 pixels.
//   * LDS-STAGED, COMPACTED SPLAT LISTS: each lane gathers one splat of the tile's sorted
//     list (flatten_ids -> means2d / conics / opacities / colours) and converts it to what the
//     inner loop wants: the conic pre-scaled by -log2(e)/2 so that
//         alpha = exp2(dx (a' dx + b' dy) + c' dy^2 + log2 o)      (5 FMA + v_exp_f32).
//     While a lane holds its splat it computes the axis-aligned extent of the
//     { alpha >= 1/255 } ellipse (half extents sqrt(2 ln(255 o) Sigma_xx), Sigma = conic^-1) and
//     tests it against the tile's live pixel rectangle; a __ballot + mbcnt prefix compacts the
//     survivors, in order, into the LDS record array.  Splats whose 3-sigma bbox touches the
//     tile but whose visible ellipse does not are never looked at again.  Culling is
//     conservative (margins below), hence exact: a culled splat has alpha < 1/255 at every
//     pixel of the tile, which the reference skips too.  The next batch's gathers are issued
//     before the current batch is processed (software prefetch), and inside a batch the next
//     record is read while the current one is evaluated.
//   * The longest lists bound the critical path (their waves run alone at the end of the kernel), so those
//     waves run at raised priority (s_setprio) and, in the tile forward, without per-batch barriers (SOLO).
//   * Backward, per (pixel, splat) with the running transmittance T and
//         D = sum_k colour_k v_out_k,   B = sum_{splats behind} fac D   (a scalar),
//     v_alpha = D T + (T_final (v_alpha_out - bg . v_out) - B) / (1 - alpha) -- the reference's
//     per-channel "buffer" vector (rasterize_to_pixels_bwd.cu:203-241) collapses to the scalar
//     B, so the per-pixel state does not grow with the channel count.  Every lane accumulates
//     over its pixels the moments
//       S0 = sum v_sigma, Sx = sum v_sigma dx, Sy, Sxx, Sxy, Syy   and   C_k = sum fac v_out_k;
//     one DPP reduction per value per (tile, splat), then
//       v_xy = (a Sx + b Sy, b Sx + c Sy), v_conic = (Sxx/2, Sxy, Syy/2), v_opacity = -S0 / o,
//     v_colour = C: algebraically the reference's formulas (bwd.cu:221-236) with the per-pixel
//     conic products hoisted out of the pixel loop.
//
// Channel counts: 1..4 use 4 pixels per lane with colours in the LDS record; 5..32 use one
// quadrant per wave (NQ = 1) with colours read from global memory at wave-uniform addresses;
// more than 32 channels: forward in exact chunks of 32, backward in ONE pass of the generic
// kernel (v_out re-read per splat) so that v_alpha -- and therefore absgrad -- sees every
// channel, as the reference's single CDIM-templated kernel does.
#include "gs_common.h"
#include "rasterize_common.h"
#include "dpp_reduce.h"

#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "rasterize_dev.h"

namespace {

// ---------------------------------------------------------------------------
// forward
//   NQ   : pixels per lane (4: whole tile, one pixel per quadrant; 1: quadrant = blockIdx.y)
//   CDIM : channels of this launch; COLOR_LDS: colours travel in the LDS record (CDIM <= 4)
// record: R0 = (mx, my, a', b')  R1 = (c', log2 o, col0, col1)  R2 = (col2, col3, idx, g)
// ---------------------------------------------------------------------------
// CKPT: write per-pixel checkpoints (T, accumulated colour) "before list entry b" for every
// b = k * seg strictly inside the tile's range, planar: ckpt[k][c][256] with c = 0 (T), 1..CDIM.
template <int NQ, int CDIM, bool COLOR_LDS, bool CKPT>
__global__ void __launch_bounds__(GS_WAVE) raster_wave_fwd_kernel(RasterArgs a, uint32_t cnt, uint32_t ch_off, float *__restrict__ ckpt, int32_t seg) {
    constexpr int REC = 3;
    __shared__ float4 s_rec[(GS_WAVE + 1) * REC];
    const uint32_t lane = threadIdx.x;
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    const uint32_t vitem = xcd_remap(blockIdx.x, gridDim.x, a.xcd_group);
    const TileGeom tg = tile_geom(a, (NQ == 4) ? vitem : (vitem >> 2));
    const uint32_t q_first = (NQ == 4) ? 0u : (vitem & 3u);
    const float *bg = a.backgrounds ? a.backgrounds + (size_t)tg.cam * a.channels + ch_off : nullptr;

    bool inside[NQ];
    float px[NQ], py[NQ];
    size_t pix[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const uint32_t q = q_first + i;
        const uint32_t ox = lx + 8u * (q & 1u), oy = ly + 8u * (q >> 1);
        const uint32_t x = tg.px0 + ox, y = tg.py0 + oy;
        inside[i] = ox < a.tile_size && oy < a.tile_size && x < a.image_width && y < a.image_height;
        px[i] = (float)x + 0.5f;
        py[i] = (float)y + 0.5f;
        pix[i] = ((size_t)tg.cam * a.image_height + y) * a.image_width + x;
    }

    if (a.masks != nullptr && !a.masks[tg.lin]) {
#pragma unroll
        for (int i = 0; i < NQ; ++i)
            if (inside[i])
                for (uint32_t k = 0; k < cnt; ++k) a.render_colors[pix[i] * a.channels + ch_off + k] = bg ? bg[k] : 0.f;
        return;
    }

    float T[NQ], Tkeep[NQ], out[NQ][CDIM];
    int32_t cur[NQ];
    bool done[NQ];
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        T[i] = 1.f;
        Tkeep[i] = 1.f;
        cur[i] = 0;
        done[i] = !inside[i];
#pragma unroll
        for (int k = 0; k < CDIM; ++k) out[i][k] = 0.f;
    }

    const Rect rect = wave_rect<NQ>(a, tg, q_first);
    const int32_t n = rect.empty ? 0 : tg.range_end - tg.range_start;
    // Batches are aligned to multiples of 64 of the GLOBAL list index (the first one is partial), so
    // that a checkpoint boundary (a multiple of seg, itself a multiple of 64) always coincides with
    // a batch start: no per-record boundary test inside the hot loop.
    const int32_t base0 = tg.range_start & ~(GS_WAVE - 1);
    const int32_t num_batches = n > 0 ? (tg.range_end - base0 + GS_WAVE - 1) / GS_WAVE : 0;
    if (n >= 4 * HEAVY_TILE) __builtin_amdgcn_s_setprio(3);
    else if (n >= 2 * HEAVY_TILE) __builtin_amdgcn_s_setprio(2);
    else if (n >= HEAVY_TILE) __builtin_amdgcn_s_setprio(1);
    auto in_range = [&](int32_t idx) { return idx >= tg.range_start && idx < tg.range_end && n > 0; };

    SplatRaw nxt = gather_splat(a, base0 + (int32_t)lane, in_range(base0 + (int32_t)lane));
    float ncol[COLOR_LDS ? CDIM : 1];
    if (COLOR_LDS) {
#pragma unroll
        for (int k = 0; k < CDIM; ++k)
            ncol[k] = (in_range(base0 + (int32_t)lane) && (uint32_t)k < cnt) ? a.colors[(size_t)nxt.g * a.s_color + ch_off + k] : 0.f;
    }
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));
    // next checkpoint boundary strictly inside (range_start, range_end), and its slot
    int32_t next_b = CKPT ? (tg.range_start / seg + 1) * seg : 0x7fffffff;
    int32_t next_k = CKPT ? next_b / seg : 0;
    auto store_ckpt = [&]() {
        float *base = ckpt + (size_t)next_k * (CDIM + 1) * 256;
#pragma unroll
        for (int i = 0; i < NQ; ++i) {
            const uint32_t p = (q_first + i) * 64u + lane;
            base[p] = done[i] ? Tkeep[i] : T[i];
#pragma unroll
            for (int k = 0; k < CDIM; ++k) base[(k + 1) * 256 + p] = out[i][k];
        }
    };

    for (int32_t b = 0; b < num_batches; ++b) {
        const int32_t batch_start = base0 + b * GS_WAVE;
        if (CKPT && batch_start == next_b && next_b < tg.range_end) { // state before list entry next_b
            store_ckpt();
            next_b += seg;
            next_k += 1;
        }
        // ---- cull + compact the prefetched splats into LDS
        SplatRaw s = nxt;
        CullSplat cs;
        const bool have = in_range(batch_start + (int32_t)lane);
        const bool live = have && cull_prepare(s, cs) && rect_touch(s, cs, rect.x0, rect.x1, rect.y0, rect.y1);
        const unsigned long long lm = __ballot(live);
        const int count = __popcll(lm);
        if (live) {
            const int slot = __popcll(lm & lt_mask);
            float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
            if (COLOR_LDS) {
                c0 = ncol[0];
                if (CDIM > 1) c1 = ncol[CDIM > 1 ? 1 : 0];
                if (CDIM > 2) c2 = ncol[CDIM > 2 ? 2 : 0];
                if (CDIM > 3) c3 = ncol[CDIM > 3 ? 3 : 0];
            }
            s_rec[slot * REC + 0] = make_float4(s.mx, s.my, -0.5f * LOG2E * s.ca, -LOG2E * s.cb);
            s_rec[slot * REC + 1] = make_float4(-0.5f * LOG2E * s.cc, __log2f(s.opac), c0, c1);
            s_rec[slot * REC + 2] = make_float4(c2, c3, __int_as_float(batch_start + (int32_t)lane), __int_as_float(s.g));
        }
        // ---- prefetch the next batch
        if (b + 1 < num_batches) {
            const int32_t ni = batch_start + GS_WAVE + (int32_t)lane;
            nxt = gather_splat(a, ni, in_range(ni));
            if (COLOR_LDS) {
#pragma unroll
                for (int k = 0; k < CDIM; ++k)
                    ncol[k] = (in_range(ni) && (uint32_t)k < cnt) ? a.colors[(size_t)nxt.g * a.s_color + ch_off + k] : 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();

        // ---- walk the compacted records, two per iteration.  The only loop-carried dependency
        // is ONE fma per record (T <- T - T a): a finished pixel keeps multiplying (its
        // contributions are masked by `done`, its final transmittance is parked in Tkeep), so the
        // alpha evaluation of the next records overlaps the composite of the current one.
        auto composite = [&](const float4 &c0, const float4 &c1, const float4 &c2, bool rec_ok) {
            float col[CDIM];
            if (COLOR_LDS) {
                col[0] = c1.z;
                if (CDIM > 1) col[CDIM > 1 ? 1 : 0] = c1.w;
                if (CDIM > 2) col[CDIM > 2 ? 2 : 0] = c2.x;
                if (CDIM > 3) col[CDIM > 3 ? 3 : 0] = c2.y;
            } else {
                const float *cp = a.colors + (size_t)__float_as_int(c2.w) * a.s_color + ch_off;
#pragma unroll
                for (int k = 0; k < CDIM; ++k) col[k] = ((uint32_t)k < cnt && rec_ok) ? cp[k] : 0.f;
            }
            const int32_t idx = __float_as_int(c2.z);
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const float dx = c0.x - px[i], dy = c0.y - py[i];
                // the same arithmetic as the tile kernels (exponent with log2(opacity) in its last fma, "sigma < 0" read as
                // pl > log2(opacity)): every compositing kernel takes bit-identical accept decisions
                const float pl = __builtin_fmaf(dx, __builtin_fmaf(c0.w, dy, c0.z * dx), __builtin_fmaf(c1.x * dy, dy, c1.y));
                const float alpha = fminf(0.999f, __builtin_amdgcn_exp2f(pl));
                const bool ok = rec_ok && !(pl > c1.y) && (alpha >= ALPHA_MIN);
                const float a_eff = ok ? alpha : 0.f;
                const float Tj = T[i];
                const float next_T = Tj - Tj * a_eff;         // the loop-carried chain
                const bool stop = ok && (next_T <= 1e-4f);    // exclusive stop
                const bool live = !done[i] && !stop;          // this record is composited
                Tkeep[i] = done[i] ? Tkeep[i] : Tj;           // transmittance before the stopping splat
                const float vis = live ? a_eff * Tj : 0.f;
#pragma unroll
                for (int k = 0; k < CDIM; ++k) out[i][k] += col[k] * vis;
                cur[i] = (live && ok) ? idx : cur[i];
                done[i] = done[i] || stop;
                T[i] = next_T;
            }
        };
        {
            int j = 0;
            for (; j + 1 < count; j += 2) {
                const float4 a0 = s_rec[j * REC + 0], a1 = s_rec[j * REC + 1], a2 = s_rec[j * REC + 2];
                const float4 b0 = s_rec[j * REC + 3], b1 = s_rec[j * REC + 4], b2 = s_rec[j * REC + 5];
                composite(a0, a1, a2, true);
                composite(b0, b1, b2, true);
            }
            if (j < count) {
                const float4 a0 = s_rec[j * REC + 0], a1 = s_rec[j * REC + 1], a2 = s_rec[j * REC + 2];
                composite(a0, a1, a2, true);
            }
        }
        // ---- early exit when every pixel of the wave is finished
        bool all_done = true;
#pragma unroll
        for (int i = 0; i < NQ; ++i) all_done = all_done && done[i];
        if (__all(all_done)) break;
        __builtin_amdgcn_wave_barrier();
    }

    if (CKPT && !rect.empty) {
        // boundaries after the last live record (or after an early exit) carry the final state
        while (next_b < tg.range_end) {
            store_ckpt();
            next_b += seg;
            next_k += 1;
        }
    }
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        if (!inside[i]) continue;
        const float Tf = done[i] ? Tkeep[i] : T[i];
        a.render_alphas[pix[i]] = 1.f - Tf;
#pragma unroll
        for (int k = 0; k < CDIM; ++k)
            if ((uint32_t)k < cnt)
                a.render_colors[pix[i] * a.channels + ch_off + k] = bg ? out[i][k] + Tf * bg[k] : out[i][k];
        a.last_ids[pix[i]] = cur[i];
    }
}

// ---------------------------------------------------------------------------
// Forward for 1..4 channels (the hot case, default): ONE 256-THREAD WORKGROUP PER TILE, wave q
// composites quadrant q, and the four waves STAGE COOPERATIVELY: per batch of 256 list entries
// every thread gathers ONE entry, tests it exactly against all four quadrant rectangles
// (rect_touch) and writes its record once; four ballots per wave give one 64-bit mask per
// (64-entry sub-batch, quadrant).  After one barrier, wave q walks the set bits of "its" four
// masks (s_ff1 on SGPRs) and reads the records as LDS broadcasts.
// Why (measured with per-wave timestamps): with one independent wave per
// quadrant the kernel's duration was the life of the heaviest waves (list 5-6k entries, 311 us of
// a 320 us kernel), and ~60% of that was the exposed latency of the two dependent gathers
// (flatten_ids -> splat) once per 64 entries, because culling leaves only ~15 records to
// evaluate per batch.  Here a heavy tile takes 4x fewer, 4x larger batches, the gathers are
// prefetched two levels deep (ids two batches ahead, splat data one batch ahead) and are hidden
// behind ~60 record evaluations, and every entry is gathered once per tile instead of four times.
// LDS is double-buffered so that one __syncthreads per batch suffices.
// Sub-batches are aligned to multiples of 64 of the GLOBAL list index, so a checkpoint boundary
// (multiple of seg) always coincides with a sub-batch start.
// ---------------------------------------------------------------------------
// WIDE (5 <= CDIM <= 16, round 5): the same kernel for feature rendering -- the record grows by the colours
// (REC = 2 + ceil((CDIM - 2) / 4) float4: 64 B at 9 channels, 96 B at 16), the walk takes two records per iteration instead of
// four (the colour FMAs are independent work already, and four records of 6 float4 would not fit the registers), the launch
// covers channels [ch_off, ch_off + cnt) of a.channels (cnt <= CDIM; 17..32 channels = two launches over halves: render_alphas /
// last_ids / costs / the T plane of the checkpoints are written by the first), checkpoints are [k][1 + a.channels][256].
#ifndef GS_WIDE_FWD_WAVES
#define GS_WIDE_FWD_WAVES 4 // waves per SIMD the 12- / 16-channel instances are held to (A/B at 16 channels: 1 = 419 us, 4 = 406, 5 = 480 with spills)
#endif
template <int CDIM, bool CKPT>
__global__ void __launch_bounds__(256, CDIM > 9 ? GS_WIDE_FWD_WAVES : 1) raster_tile_fwd_kernel(RasterArgs a, float *__restrict__ ckpt, int32_t seg, int32_t solo_min,
                                                              uint32_t *__restrict__ cost_head, uint32_t *__restrict__ cost_body,
                                                              uint32_t *__restrict__ body_tile, uint32_t *__restrict__ class_count,
                                                              ZeroFill zf, uint32_t ch_off, uint32_t cnt) {
    constexpr bool WIDE = CDIM > 4;
    constexpr int NCW = WIDE ? (CDIM - 2 + 3) / 4 : 0; // float4s of colours 2.. behind R0, R1
    constexpr int REC = WIDE ? 2 + NCW : 3;
    constexpr int BATCH = 256;
    // record buffers: 2 (staging of batch b + 1 overlaps the compositing of batch b across the four waves: one barrier per batch);
    // the 12- and 16-channel instances take ONE (half the LDS -- 27 instead of 54 KB at 16 channels, i.e. no longer 3 workgroups
    // per CU -- for a second barrier per batch: 537 -> 415 us at 16 channels; at 9 channels a wash, 275 vs 281)
    constexpr int NBUF = (WIDE && CDIM > 9) ? 1 : 2;
    static_assert((NBUF * BATCH * REC + REC) * 16 < 65536, "record offsets are 16-bit");
    // records of both buffers in ONE array + a null record (alpha = 0) that pads every list to a multiple of four
    __shared__ float4 s_rec[NBUF * BATCH * REC + REC];
    constexpr uint32_t NULL_REC_OFF = (uint32_t)NBUF * BATCH * REC * 16u; // byte offset of the null record
    // per (buffer, sub-batch, quadrant): byte offsets (into s_rec) of the records that touch the quadrant, in list order
    typedef uint16_t list_t; // 16-bit entries keep the workgroup under 32 KB of LDS (5 per CU); 32-bit ones measured 0.252 vs 0.244 ms
    __shared__ __attribute__((aligned(16))) list_t s_list[NBUF][4][4][72];
    __shared__ unsigned long long s_mask[NBUF][4][4]; // [buffer][sub-batch (= staging wave)][quadrant]
    __shared__ uint32_t s_done[NBUF][4];              // [buffer][quadrant]
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint32_t w = __builtin_amdgcn_readfirstlane(tid >> 6); // staging sub-batch AND composited quadrant
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    if (tid < REC) s_rec[NBUF * BATCH * REC + tid] = make_float4(0.f, tid == 1 ? -__builtin_inff() : 0.f, 0.f, 0.f); // log2(opacity) = -inf
    // the class counters of the backward's work list (seg_items_build_kernel runs after this kernel, in the same call)
    // (a chunked forward -- 17..32 channels as two launches -- writes the counters and the costs in its FIRST launch only: culling does
    // not depend on the channel, and the second launch would only repeat the same stores)
    const bool cost_owner = !WIDE || ch_off == 0u;
    if (CKPT && cost_owner && blockIdx.x == 0 && tid < (uint32_t)COST_CLASSES) class_count[tid] = 0u;
    const TileGeom tg = tile_geom(a, a.tile_order != nullptr ? a.tile_order[blockIdx.x] : xcd_remap(blockIdx.x, gridDim.x, a.xcd_group));
    const float *bg = a.backgrounds ? a.backgrounds + (size_t)tg.cam * a.channels + (WIDE ? ch_off : 0u) : nullptr;
    const uint32_t CH = WIDE ? a.channels : (uint32_t)CDIM; // channels of the output image / checkpoint planes

    const uint32_t ox = lx + 8u * (w & 1u), oy = ly + 8u * (w >> 1);
    const uint32_t x = tg.px0 + ox, y = tg.py0 + oy;
    const bool inside = ox < a.tile_size && oy < a.tile_size && x < a.image_width && y < a.image_height;
    const float px = (float)x + 0.5f, py = (float)y + 0.5f;
    const size_t pix = ((size_t)tg.cam * a.image_height + y) * a.image_width + x;

    // Side job: this workgroup's slice of the buffer the matching backward will accumulate into (zf, see gs_rasterize_fwd).
    // Issued as the workgroup's LAST instructions: nothing waits for the stores, and the memory pipes are mostly idle
    // while the chip composites -- the separate 64 MB fill kernel (+ its launch gap) of the backward disappears.
    // (Issued right after the first gathers instead: 201 -> 225 us with a 300 MB job, round 3.)
    auto zero_fill_slice = [&]() {
        if (!CKPT || zf.ptr == nullptr) return;
        const size_t base = (size_t)blockIdx.x * zf.per_block;
        for (uint32_t i = tid; i < zf.per_block; i += 256u)
            if (base + i < zf.n) { // non-temporal: the zeros must not push the splat rows / checkpoints out of the memory-side cache
                typedef float v4f __attribute__((ext_vector_type(4)));
                __builtin_nontemporal_store((v4f){0.f, 0.f, 0.f, 0.f}, reinterpret_cast<v4f *>(zf.ptr + base + i));
            }
    };

    if (a.masks != nullptr && !a.masks[tg.lin]) {
        if (inside) {
#pragma unroll
            for (int k = 0; k < CDIM; ++k)
                if (!WIDE || (uint32_t)k < cnt) a.render_colors[pix * CH + (WIDE ? ch_off : 0u) + k] = bg ? bg[k] : 0.f;
        }
        zero_fill_slice();
        return;
    }

    float qx0[4], qx1[4], qy0[4], qy1[4];
    bool qdone[4]; // block-uniform: quadrant finished (or empty)
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const Rect r = wave_rect<1>(a, tg, (uint32_t)q);
        qx0[q] = r.x0; qx1[q] = r.x1; qy0[q] = r.y0; qy1[q] = r.y1;
        qdone[q] = r.empty;
    }

    float T = 1.f, out[CDIM]; // T freezes at the stopping splat: it is the transmittance in front of it
    int32_t cur = 0;
    bool done = !inside;
#pragma unroll
    for (int k = 0; k < CDIM; ++k) out[k] = 0.f;

    const int32_t n = tg.range_end - tg.range_start;
    const int32_t base0 = tg.range_start & ~(GS_WAVE - 1);
    const int32_t num_batches = n > 0 ? (tg.range_end - base0 + BATCH - 1) / BATCH : 0;
    // graded issue priority: the longest lists bound the kernel, they must win arbitration on their SIMD
    if (n >= 4 * HEAVY_TILE) __builtin_amdgcn_s_setprio(3);
    else if (n >= 2 * HEAVY_TILE) __builtin_amdgcn_s_setprio(2);
    else if (n >= HEAVY_TILE) __builtin_amdgcn_s_setprio(1);
    auto in_range = [&](int32_t idx) { return idx >= tg.range_start && idx < tg.range_end; };
    auto load_id = [&](int32_t idx) { return in_range(idx) ? a.flatten_ids[idx] : -1; };
    struct Staged {
        SplatRaw s;
        float col[CDIM];
    };
    auto gather = [&](int32_t g, Staged &o) {
        o.s.g = 0;
        o.s.mx = o.s.my = o.s.ca = o.s.cb = o.s.cc = o.s.opac = 0.f;
#pragma unroll
        for (int k = 0; k < CDIM; ++k) o.col[k] = 0.f;
        if (g >= 0) {
            if constexpr (WIDE) fetch_splat_wide<CDIM>(a, g, o.s, o.col, ch_off, cnt);
            else fetch_splat<CDIM>(a, g, o.s, o.col);
        }
    };
    // the record of one staged entry (R0, R1 and the colour float4s behind them)
    auto write_rec = [&](float4 *r, const Staged &st) {
        r[0] = make_float4(st.s.mx, st.s.my, -0.5f * LOG2E * st.s.ca, -LOG2E * st.s.cb);
        r[1] = make_float4(-0.5f * LOG2E * st.s.cc, __log2f(st.s.opac), st.col[0], st.col[CDIM > 1 ? 1 : 0]);
#pragma unroll
        for (int j = 0; j < NCW; ++j) {
            auto c = [&](int k) { return k < CDIM ? st.col[k < CDIM ? k : 0] : 0.f; };
            r[2 + j] = make_float4(c(2 + 4 * j), c(3 + 4 * j), c(4 + 4 * j), c(5 + 4 * j));
        }
    };
    int32_t id_cur = load_id(base0 + (int32_t)tid);
    int32_t id_nxt = load_id(base0 + BATCH + (int32_t)tid);
    Staged nxt;
    gather(id_cur, nxt);

    // first boundary this workgroup stores: strictly inside the list
    int32_t next_b = CKPT ? (tg.range_start / seg + 1) * seg : 0x7fffffff;
    int32_t next_k = CKPT ? next_b / seg : 0;
    // COST of every backward work item (tile, segment) = records this wave evaluated inside the segment: the backward's
    // work list is ordered by it, longest first (seg_items_sorted_kernel).  The tile's first segment reports into
    // cost_head[tile][wave]; a later segment k owns boundary k * seg and reports into cost_body[k][wave] -- written once
    // per (item, wave), no zero-fill and no atomics.
    uint32_t evals = 0;
    bool first_seg = true;
    auto store_cost = [&]() { // the segment in front of boundary next_k has ended
        if (lane == 0 && cost_owner) {
            if (first_seg) cost_head[tg.lin * 4u + w] = evals;
            else {
                cost_body[(size_t)(next_k - 1) * 4u + w] = evals;
                if (w == 0u) body_tile[next_k - 1] = tg.lin; // the tile that owns boundary (next_k - 1) * seg
            }
        }
        evals = 0;
        first_seg = false;
    };
    auto store_ckpt = [&]() {
        float *base = ckpt + (size_t)next_k * (CH + 1) * 256;
        const uint32_t p = w * 64u + lane;
        if (!WIDE || ch_off == 0u) base[p] = T;
#pragma unroll
        for (int k = 0; k < CDIM; ++k)
            if (!WIDE || (uint32_t)k < cnt) base[((WIDE ? ch_off : 0u) + k + 1) * 256 + p] = out[k];
        store_cost();
    };

    // ---- the record walk of one (sub-batch, quadrant) list, shared by the cooperative and the solo path
#ifndef GS_WIDE_GW9
#define GS_WIDE_GW9 2
#endif
    constexpr int GW = WIDE ? (CDIM <= 9 ? GS_WIDE_GW9 : 2) : 4; // records per iteration of the bulk loop
    struct alignas(sizeof(list_t) * GW) Pack { list_t v[GW]; };
    // G records in one basic block (G = 4: the bulk of a list; G = 1: its last 1-3 records -- the lists used to be walked as
    // whole groups of four with their null-record padding, ~1.5 wasted evaluations per (sub-batch, quadrant) list = 9 % of
    // all evaluations; the padding is still written (harmless) but no longer walked)
    auto group = [&](auto gtag, const list_t *offs, uint32_t &cur_off) {
        constexpr int G = decltype(gtag)::value;
                float4 c0[G], c1[G];
                float c2x[G], c2y[G]; // colours 2, 3 (only what CDIM needs is read)
                float4 cw[G][WIDE ? NCW : 1]; // WIDE: colours 2 .. CDIM - 1
                uint32_t off[G];
                {
#pragma unroll
                    for (int g = 0; g < G; ++g) off[g] = offs[g];
#pragma unroll
                    for (int g = 0; g < G; ++g) {
                        const float4 *r = reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(s_rec) + off[g]);
                        c0[g] = r[0];
                        c1[g] = r[1];
                        c2x[g] = c2y[g] = 0.f;
                        if constexpr (WIDE) {
#pragma unroll
                            for (int j = 0; j < NCW; ++j) cw[g][j] = r[2 + j];
                        }
                        if (CDIM == 3) c2x[g] = reinterpret_cast<const float *>(r + 2)[0];
                        if (CDIM == 4) {
                            const float2 v = reinterpret_cast<const float2 *>(r + 2)[0];
                            c2x[g] = v.x;
                            c2y[g] = v.y;
                        }
                    }
                }
                // The forward is ~80 % VALU-issue bound (every extra instruction per record costs ~6 us of the kernel, round 2):
                // the selects are merged (one for the alpha actually applied, one for the last contributor, both on the same
                // condition) and the conditions themselves combined on the scalar unit.
                float alpha[G];
                bool ok[G];
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const float dx = c0[g].x - px, dy = c0[g].y - py;
                    // pl = power + log2(opacity), power = -sigma log2(e) = dx (a' dx + b' dy) + c' dy^2; the constant rides in
                    // the last fma, and "sigma < 0" (reject) is read off as pl > log2(opacity): the two differ only for
                    // |sigma| below the rounding of pl, where the sign of a computed sigma is rounding noise in any case
                    const float lo = c1[g].y;
                    const float pl = __builtin_fmaf(dx, __builtin_fmaf(c0[g].w, dy, c0[g].z * dx), __builtin_fmaf(c1[g].x * dy, dy, lo));
                    alpha[g] = fminf(0.999f, __builtin_amdgcn_exp2f(pl));
                    ok[g] = !(pl > lo) && (alpha[g] >= ALPHA_MIN);
                }
                // T is FROZEN at the stopping splat (= the transmittance in front of it, what the epilogue and the
                // checkpoints need), so a live pixel always has T > 1e-4 and a rejected record (a_eff = 0) cannot stop it
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    // exclusive stop: the record that would take T to <= 1e-4 is not composited and ends the pixel.  ONE
                    // select (the alpha actually applied: 0 for a finished or finishing pixel) instead of separate selects
                    // for the weight and for T -- T is then recomputed with it (a second fma is cheaper than a select)
                    const float Tj = T;
                    const bool stop = ok[g] && (__builtin_fmaf(-Tj, alpha[g], Tj) <= 1e-4f);
                    done = done || stop;
                    const bool use = ok[g] && !done; // composited: accepted, and the pixel neither finished nor finishing
                    const float a_use = use ? alpha[g] : 0.f;
                    const float vis = a_use * Tj;
                    out[0] += c1[g].z * vis;
                    if (CDIM > 1) out[CDIM > 1 ? 1 : 0] += c1[g].w * vis;
                    if constexpr (WIDE) {
#pragma unroll
                        for (int k = 2; k < CDIM; ++k) {
                            const float4 v = cw[g][(k - 2) / 4];
                            const int e = (k - 2) % 4;
                            out[k] += (e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w) * vis;
                        }
                    } else {
                    if (CDIM > 2) out[CDIM > 2 ? 2 : 0] += c2x[g] * vis;
                    if (CDIM > 3) out[CDIM > 3 ? 3 : 0] += c2y[g] * vis;
                    }
                    cur_off = use ? off[g] : cur_off;
                    T = __builtin_fmaf(-Tj, a_use, Tj);
                }
    };
    auto walk = [&](const list_t *lst, const uint32_t cnt, uint32_t &cur_off) {
            // FOUR records per iteration, in one basic block: a lone wave issues one instruction per
            // ~5 cycles and a record is a ~25-deep dependent chain, so one record at a time costs
            // ~390 cycles (measured) against ~40 instructions x 5 cycles of issue.  The four alpha
            // evaluations are independent; only T <- T - T a (one fma) is carried from record to record.
            // The records come from the compacted list of this (sub-batch, quadrant): no bit scan, no validity flags.
            // (No explicit prefetch of the next group: it cost 40 VGPRs, i.e. two waves per SIMD, and the
            // kernel's duration is rounds of workgroups x their lifetime, not the walk of one list.)
            const uint32_t full = cnt & ~(uint32_t)(GW - 1);
            Pack pk_next = *reinterpret_cast<const Pack *>(lst); // GW record offsets, one broadcast read
            for (uint32_t j = 0; j < full; j += GW) {
                const Pack pk = pk_next;
                pk_next = *reinterpret_cast<const Pack *>(lst + j + GW); // next group's offsets (row is 72 long: in bounds)
                group(std::integral_constant<int, GW>{}, pk.v, cur_off);
            }
            for (uint32_t j = full; j < cnt; ++j) group(std::integral_constant<int, 1>{}, lst + j, cur_off);
    };

    // ---- SOLO path for the longest lists.  A tile's lifetime under the cooperative scheme is the sum over its batches
    // of the BUSIEST quadrant's work (the four waves meet at a barrier every 256 entries); the tiles with thousands of
    // entries run alone at the end of the kernel and set its duration.  There every wave walks the whole list by itself,
    // 64 entries at a time, culling against its own quadrant only: no barrier, its time is its own work, and a quadrant
    // that is done leaves.  (4x the gathers for these tiles -- 18 % of the pairs at config 2.)
    const bool solo = solo_min > 0 && n >= solo_min;
    if (solo) {
        float rx0 = qx0[0], ry0 = qy0[0];
#pragma unroll
        for (int q = 1; q < 4; ++q)
            if (w == (uint32_t)q) { rx0 = qx0[q]; ry0 = qy0[q]; }
        const int32_t n_sb = (tg.range_end - base0 + GS_WAVE - 1) / GS_WAVE;
        int32_t sid_cur = load_id(base0 + (int32_t)lane);
        int32_t sid_nxt = load_id(base0 + GS_WAVE + (int32_t)lane);
        Staged snx;
        gather(sid_cur, snx);
        __syncthreads(); // the null record (written by the first lanes of wave 0) is visible to every wave
        for (int32_t sbi = 0; sbi < n_sb; ++sbi) {
            const uint32_t buf = (uint32_t)sbi & (uint32_t)(NBUF - 1);
            const int32_t sb_start = base0 + sbi * GS_WAVE;
            const uint32_t slot = buf * BATCH + w * GS_WAVE + lane; // my record slot: the wave's own quarter of the buffer
            unsigned long long m;
            {
                const Staged st = snx;
                CullSplat cs;
                const bool live = sid_cur >= 0 && cull_prepare(st.s, cs);
                // cull against the bounding rectangle of the pixels that are still being composited (finished pixels ignore
                // every further record): the longest lists are walked to their end by a few unsaturated pixels
                const unsigned long long alive = __ballot(!done);
                bool touch = false;
                if (alive != 0ull) {
                    const LiveRect lr = live_rect(alive, rx0, ry0);
                    touch = live && rect_touch(st.s, cs, lr.x0, lr.x1, lr.y0, lr.y1);
                }
                m = __ballot(touch);
                if (touch) {
                    float4 *r = &s_rec[slot * REC];
                    if constexpr (WIDE) write_rec(r, st);
                    else {
                    float c0 = st.col[0], c1 = 0.f, c2 = 0.f, c3 = 0.f;
                    if (CDIM > 1) c1 = st.col[CDIM > 1 ? 1 : 0];
                    if (CDIM > 2) c2 = st.col[CDIM > 2 ? 2 : 0];
                    if (CDIM > 3) c3 = st.col[CDIM > 3 ? 3 : 0];
                    r[0] = make_float4(st.s.mx, st.s.my, -0.5f * LOG2E * st.s.ca, -LOG2E * st.s.cb);
                    r[1] = make_float4(-0.5f * LOG2E * st.s.cc, __log2f(st.s.opac), c0, c1);
                    if (CDIM > 2) r[2] = make_float4(c2, c3, 0.f, 0.f);
                    }
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
                    s_list[buf][w][w][below] = (list_t)(slot * REC * 16u);
                }
                const uint32_t cnt = (uint32_t)__popcll(m);
                if (lane < 3u && cnt + lane < ((cnt + 3u) & ~3u)) s_list[buf][w][w][cnt + lane] = (list_t)NULL_REC_OFF;
            }
            sid_cur = sid_nxt;
            if (sbi + 1 < n_sb) gather(sid_cur, snx);
            sid_nxt = (sbi + 2 < n_sb) ? load_id(sb_start + 2 * GS_WAVE + (int32_t)lane) : -1;
            __builtin_amdgcn_wave_barrier(); // (LDS operations of one wave complete in order)
            if (CKPT && sb_start == next_b && next_b < tg.range_end) { // state before list entry next_b
                store_ckpt();
                next_b += seg;
                next_k += 1;
            }
            if (m == 0ull) continue;
            uint32_t cur_off = 0xffffffffu;
            if (CKPT) evals += (uint32_t)__popcll(m);
            walk(&s_list[buf][w][w][0], (uint32_t)__popcll(m), cur_off);
            if (cur_off != 0xffffffffu)
                cur = sb_start + (int32_t)((WIDE ? (cur_off >> 4) / (uint32_t)REC : (((cur_off >> 4) * 43691u) >> 17)) & 63u);
            if (__all(done)) break;
        }
    } else
    for (int32_t b = 0; b < num_batches; ++b) {
        const uint32_t buf = (uint32_t)b & (uint32_t)(NBUF - 1);
        const int32_t batch_start = base0 + b * BATCH;
        // ---- stage: exact cull of my entry against the four quadrants
        {
            const Staged st = nxt;
            const bool have = id_cur >= 0;
            CullSplat cs;
            const bool live = have && cull_prepare(st.s, cs);
            unsigned long long m[4];
            bool any_touch = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // (culling against the rectangle of the still-unfinished pixels, as the solo path does, was measured here too:
                // 0.210 -> 0.215 ms -- these tiles are short, the masks reach the staging threads two batches late)
                const bool touch = live && !qdone[q] && rect_touch(st.s, cs, qx0[q], qx1[q], qy0[q], qy1[q]);
                any_touch |= touch;
                m[q] = __ballot(touch);
            }
            if (any_touch) {
                float4 *r = &s_rec[(buf * BATCH + tid) * REC];
                if constexpr (WIDE) write_rec(r, st);
                else {
                float c0 = st.col[0], c1 = 0.f, c2 = 0.f, c3 = 0.f;
                if (CDIM > 1) c1 = st.col[CDIM > 1 ? 1 : 0];
                if (CDIM > 2) c2 = st.col[CDIM > 2 ? 2 : 0];
                if (CDIM > 3) c3 = st.col[CDIM > 3 ? 3 : 0];
                r[0] = make_float4(st.s.mx, st.s.my, -0.5f * LOG2E * st.s.ca, -LOG2E * st.s.cb);
                r[1] = make_float4(-0.5f * LOG2E * st.s.cc, __log2f(st.s.opac), c0, c1);
                if (CDIM > 2) r[2] = make_float4(c2, c3, 0.f, 0.f);
                }
            }
            {
                // compacted lists: a lone wave issues one instruction per ~5 cycles WHATEVER its kind, so the consumer
                // must not spend a dozen scalar instructions per record on bit scans -- it reads ready-made offsets
                const uint32_t my_off = (buf * BATCH + tid) * REC * 16u;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi((uint32_t)(m[q] >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m[q], 0u));
                    if ((m[q] >> lane) & 1ull) s_list[buf][w][q][below] = (list_t)my_off;
                    const uint32_t cnt = (uint32_t)__popcll(m[q]);
                    if (lane < 3u && cnt + lane < ((cnt + 3u) & ~3u)) s_list[buf][w][q][cnt + lane] = (list_t)NULL_REC_OFF;
                }
            }
            const bool wave_done = __all(done); // evaluated by all 64 lanes, before the branch
            if (lane == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) s_mask[buf][w][q] = m[q];
                s_done[buf][w] = wave_done ? 1u : 0u;
            }
        }
        // ---- prefetch: splat data of the next batch (its ids are already here), ids of the one after
        id_cur = id_nxt;
        if (b + 1 < num_batches) gather(id_cur, nxt);
        id_nxt = (b + 2 < num_batches) ? load_id(batch_start + 2 * BATCH + (int32_t)tid) : -1;
        // LDS-only barrier: __syncthreads() would also drain vmcnt, i.e. wait for the gathers just issued
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        {
            const uint32_t d0 = s_done[buf][0], d1 = s_done[buf][1], d2 = s_done[buf][2], d3 = s_done[buf][3];
            if (d0 & d1 & d2 & d3) break; // every pixel of the tile is finished (block-uniform)
            qdone[0] = qdone[0] || d0; qdone[1] = qdone[1] || d1; qdone[2] = qdone[2] || d2; qdone[3] = qdone[3] || d3;
        }
        // ---- composite my quadrant over the four sub-batches
        uint32_t cur_off = 0xffffffffu; // record offset of the last contributor inside this batch (none yet)
#pragma unroll 1
        for (int sub = 0; sub < 4; ++sub) {
            const int32_t sb_start = batch_start + sub * GS_WAVE;
            if (sb_start >= tg.range_end) break;
            if (CKPT && sb_start == next_b && next_b < tg.range_end) { // state before list entry next_b
                store_ckpt();
                next_b += seg;
                next_k += 1;
            }
            unsigned long long m = s_mask[buf][sub][w];
            m = ((unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(m >> 32)) << 32) |
                (unsigned long long)(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)m); // (the builtin returns int: no sign extension)
            if (m == 0ull) continue;
            if (CKPT) evals += (uint32_t)__popcll(m);
            walk(&s_list[buf][sub][w][0], (uint32_t)__popcll(m), cur_off);
            if (__all(done)) break;
        }
        if (cur_off != 0xffffffffu) // offset -> list index: (offset / 16 - buffer base) / 3, exact for these small multiples of 3
            cur = batch_start + (int32_t)(WIDE ? ((cur_off >> 4) - buf * (uint32_t)(BATCH * REC)) / (uint32_t)REC
                                               : ((((cur_off >> 4) - buf * (uint32_t)(BATCH * REC)) * 43691u) >> 17));
        if (NBUF == 1) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // nobody restages while a wave still walks this batch
    }

    if (CKPT && n > 0) {
        // boundaries after the last composited record (or after an early exit) carry the final state
        while (next_b < tg.range_end) {
            store_ckpt();
            next_b += seg;
            next_k += 1;
        }
        store_cost(); // the tile's last segment
    }
    if (inside) {
        const float Tf = T;
        if (!WIDE || ch_off == 0u) a.render_alphas[pix] = 1.f - Tf;
#pragma unroll
        for (int k = 0; k < CDIM; ++k)
            if (!WIDE || (uint32_t)k < cnt) a.render_colors[pix * CH + (WIDE ? ch_off : 0u) + k] = bg ? out[k] + Tf * bg[k] : out[k];
        if (!WIDE || ch_off == 0u) a.last_ids[pix] = cur;
    }
    zero_fill_slice();
}

// The forward's workgroups in the order "longest list first": workgroup b composites tile order[b].  The hardware starts
// workgroups in index order, so the heavy tiles (up to ~7000 entries at BASELINE config 2 against 490 on average) begin at
// once instead of somewhere in the last rounds, where each of them used to hold the kernel's end back while the chip
// drained: raster_tile_fwd_kernel 197.5 -> 178.7 us (round 3; the same idea as the backward's cost-ordered work list).
// One workgroup: a counting sort of the tiles by list length in classes of 4 entries (2048 classes, the longest first; the
// order inside a class is whatever the atomics give -- it only decides which workgroup index a tile gets).
constexpr int ORDER_CLASSES = 2048, ORDER_THREADS = 1024, ORDER_PER = 8;
__global__ void __launch_bounds__(ORDER_THREADS) tile_order_kernel(uint32_t n_tiles_all, const int32_t *__restrict__ tile_offsets,
                                                                  uint32_t n_isects, uint32_t *__restrict__ order) {
    __shared__ uint32_t s_cnt[ORDER_CLASSES];
    __shared__ uint32_t s_wave[ORDER_THREADS / 64];
    const uint32_t tid = threadIdx.x;
    s_cnt[tid] = 0u;
    s_cnt[tid + ORDER_THREADS] = 0u;
    auto cls_of = [&](uint32_t t) {
        const uint32_t end = (t + 1u == n_tiles_all) ? n_isects : (uint32_t)tile_offsets[t + 1u];
        const uint32_t len = end - (uint32_t)tile_offsets[t];
        return (uint32_t)(ORDER_CLASSES - 1) - min(len >> 2, (uint32_t)(ORDER_CLASSES - 1)); // class 0 = the longest lists
    };
    // the first ORDER_PER x 1024 tiles keep their class in registers (all of them at 1080p: 8160 tiles)
    uint32_t cls[ORDER_PER];
#pragma unroll
    for (int k = 0; k < ORDER_PER; ++k) {
        const uint32_t t = (uint32_t)k * ORDER_THREADS + tid;
        cls[k] = t < n_tiles_all ? cls_of(t) : 0xffffffffu;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ORDER_PER; ++k)
        if (cls[k] != 0xffffffffu) atomicAdd(&s_cnt[cls[k]], 1u);
    for (uint32_t t = ORDER_PER * ORDER_THREADS + tid; t < n_tiles_all; t += ORDER_THREADS) atomicAdd(&s_cnt[cls_of(t)], 1u);
    __syncthreads();
    // exclusive scan of the class counts: two consecutive classes per thread
    const uint32_t c0 = s_cnt[2u * tid], c1 = s_cnt[2u * tid + 1u];
    uint32_t v = c0 + c1;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t u = __shfl_up(v, off, 64);
        if ((int)(tid & 63u) >= off) v += u;
    }
    if ((tid & 63u) == 63u) s_wave[tid >> 6] = v;
    __syncthreads();
    const uint32_t wv = (tid & 63u) < (uint32_t)(ORDER_THREADS / 64) ? s_wave[tid & 63u] : 0u; // the 16 wave totals, scanned by every wave
    uint32_t ws = wv;
#pragma unroll
    for (int off = 1; off < ORDER_THREADS / 64; off <<= 1) {
        const uint32_t u = __shfl_up(ws, off, 64);
        if ((int)(tid & 63u) >= off) ws += u;
    }
    const uint32_t base = __shfl(ws - wv, tid >> 6, 64); // exclusive total of the waves in front of mine
    const uint32_t excl = base + v - (c0 + c1);
    s_cnt[2u * tid] = excl; // (every thread rewrites exactly the two entries it read)
    s_cnt[2u * tid + 1u] = excl + c0;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < ORDER_PER; ++k)
        if (cls[k] != 0xffffffffu) order[atomicAdd(&s_cnt[cls[k]], 1u)] = (uint32_t)k * ORDER_THREADS + tid;
    for (uint32_t t = ORDER_PER * ORDER_THREADS + tid; t < n_tiles_all; t += ORDER_THREADS) order[atomicAdd(&s_cnt[cls_of(t)], 1u)] = t;
}

template <int CDIM>
void launch_tile_fwd(const RasterArgs &a, float *ckpt, int32_t seg, int32_t solo_min, uint32_t *cost_head, uint32_t *cost_body,
                     uint32_t *body_tile, uint32_t *class_count, ZeroFill zf, hipStream_t st, uint32_t ch_off = 0u, uint32_t cnt = (uint32_t)CDIM) {
    dim3 grid(a.C * a.tile_width * a.tile_height);
    if (ckpt != nullptr)
        hipLaunchKernelGGL((raster_tile_fwd_kernel<CDIM, true>), grid, dim3(256), 0, st, a, ckpt, seg, solo_min, cost_head, cost_body, body_tile, class_count, zf, ch_off, cnt);
    else
        hipLaunchKernelGGL((raster_tile_fwd_kernel<CDIM, false>), grid, dim3(256), 0, st, a, ckpt, seg, solo_min, cost_head, cost_body, body_tile, class_count, zf, ch_off, cnt);
}

// ---------------------------------------------------------------------------
// backward
//   CMODE: 0 = colours in the LDS record (CDIM <= 4, v_out in registers)
//          1 = colours from global, v_out in registers (CDIM <= 32)
//          2 = any channel count: colours and v_out from global per splat (CDIM unused)
// record: R0, R1 as forward; R2 = (col2, col3, idx, g); R3 = (a, b, c, o)
// ---------------------------------------------------------------------------
// SEG: blockIdx.x indexes a work item (tile, k): the list entries [k*seg, (k+1)*seg) of that
// tile.  The state at the item's far end comes from the forward's checkpoint k+1
// (T, accumulated colour) and the final render: B = v_out . (colour_final - colour_ckpt).

template <int NQ, int CDIM, int CMODE, bool ABS>
__global__ void __launch_bounds__(GS_WAVE) raster_wave_bwd_kernel(RasterArgs a, RasterGradArgs ga, uint32_t cnt, uint32_t ch_off, int use_v_alpha) {
    constexpr int REC = 4;
    constexpr int CR = (CMODE == 2) ? 1 : CDIM; // registers for v_out / colour sums
    __shared__ float4 s_rec[(GS_WAVE + 1) * REC];
    const uint32_t lane = threadIdx.x;
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    const uint32_t vitem = xcd_remap(blockIdx.x, gridDim.x, a.xcd_group);
    const TileGeom tg = tile_geom(a, (NQ == 4) ? vitem : (vitem >> 2));
    if (a.masks != nullptr && !a.masks[tg.lin]) return;
    const uint32_t q_first = (NQ == 4) ? 0u : (vitem & 3u);
    const Rect rect = wave_rect<NQ>(a, tg, q_first);
    if (rect.empty || tg.range_end <= tg.range_start) return;

    bool inside[NQ];
    float px[NQ], py[NQ], T[NQ], Tw[NQ], Bq[NQ], vc[NQ][CR];
    int32_t bin_final[NQ];
    size_t pixv[NQ];
    int64_t vpix[NQ]; // the same pixel in v_render_colors (its own strides)
    const float *bg = a.backgrounds ? a.backgrounds + (size_t)tg.cam * a.channels + ch_off : nullptr;
    int32_t bin_max = -1;
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const uint32_t q = q_first + i;
        const uint32_t ox = lx + 8u * (q & 1u), oy = ly + 8u * (q >> 1);
        const uint32_t x = tg.px0 + ox, y = tg.py0 + oy;
        inside[i] = ox < a.tile_size && oy < a.tile_size && x < a.image_width && y < a.image_height;
        px[i] = (float)x + 0.5f;
        py[i] = (float)y + 0.5f;
        const size_t pix = inside[i] ? ((size_t)tg.cam * a.image_height + y) * a.image_width + x : 0;
        pixv[i] = pix * a.channels + ch_off;
        vpix[i] = (int64_t)pix * ga.s_vrc_pix + (int64_t)ch_off * ga.s_vrc_ch;
        const float T_final = inside[i] ? 1.f - ga.render_alphas[pix] : 1.f;
        T[i] = T_final;
        Bq[i] = 0.f;
        float bg_dot = 0.f;
        if (CMODE != 2) {
#pragma unroll
            for (int k = 0; k < CR; ++k) {
                vc[i][k] = (inside[i] && (uint32_t)k < cnt) ? ga.v_render_colors[vpix[i] + k * ga.s_vrc_ch] : 0.f;
                if (bg != nullptr && (uint32_t)k < cnt) bg_dot += bg[k] * vc[i][k];
            }
        } else {
            vc[i][0] = 0.f;
            if (bg != nullptr && inside[i])
                for (uint32_t k = 0; k < cnt; ++k) bg_dot += bg[k] * ga.v_render_colors[vpix[i] + k * ga.s_vrc_ch];
        }
        const float v_a = (inside[i] && use_v_alpha) ? ga.v_render_alphas[pix] : 0.f;
        Tw[i] = T_final * (v_a - bg_dot);
        bin_final[i] = inside[i] ? ga.last_ids[pix] : -1; // never matches
        bin_max = max(bin_max, bin_final[i]);
    }
    bin_max = wave_max_i32(bin_max);
    if (bin_max < tg.range_start) return; // nothing was composited in this wave's pixels
    // nothing behind bin_max contributes: start there and walk back to front
    const int32_t first = min(tg.range_end - 1, bin_max);
    const int32_t total = first - tg.range_start + 1;
    const int32_t num_batches = (total + GS_WAVE - 1) / GS_WAVE;
    if (total >= HEAVY_TILE) __builtin_amdgcn_s_setprio(2);

    SplatRaw nxt = gather_splat(a, first - (int32_t)lane, first - (int32_t)lane >= tg.range_start);
    float ncol[CMODE == 0 ? CDIM : 1];
    if (CMODE == 0) {
#pragma unroll
        for (int k = 0; k < CDIM; ++k)
            ncol[k] = (first - (int32_t)lane >= tg.range_start && (uint32_t)k < cnt)
                          ? a.colors[(size_t)nxt.g * a.s_color + ch_off + k] : 0.f;
    }
    const unsigned long long lt_mask = (lane == 0) ? 0ull : (~0ull >> (64 - lane));

    for (int32_t b = 0; b < num_batches; ++b) {
        const int32_t batch_end = first - b * GS_WAVE; // lane l holds list index batch_end - l
        SplatRaw s = nxt;
        CullSplat cs;
        const int32_t my_idx = batch_end - (int32_t)lane;
        const bool live = (my_idx >= tg.range_start) && cull_prepare(s, cs) && rect_touch(s, cs, rect.x0, rect.x1, rect.y0, rect.y1);
        const unsigned long long lm = __ballot(live);
        const int count = __popcll(lm);
        if (live) {
            const int slot = __popcll(lm & lt_mask);
            float c0 = 0.f, c1 = 0.f, c2 = 0.f, c3 = 0.f;
            if (CMODE == 0) {
                c0 = ncol[0];
                if (CDIM > 1) c1 = ncol[CDIM > 1 ? 1 : 0];
                if (CDIM > 2) c2 = ncol[CDIM > 2 ? 2 : 0];
                if (CDIM > 3) c3 = ncol[CDIM > 3 ? 3 : 0];
            }
            s_rec[slot * REC + 0] = make_float4(s.mx, s.my, -0.5f * LOG2E * s.ca, -LOG2E * s.cb);
            s_rec[slot * REC + 1] = make_float4(-0.5f * LOG2E * s.cc, __log2f(s.opac), c0, c1);
            s_rec[slot * REC + 2] = make_float4(c2, c3, __int_as_float(my_idx), __int_as_float(s.g));
            s_rec[slot * REC + 3] = make_float4(s.ca, s.cb, s.cc, s.opac);
        }
        if (b + 1 < num_batches) {
            const int32_t ni = first - (b + 1) * GS_WAVE - (int32_t)lane;
            nxt = gather_splat(a, ni, ni >= tg.range_start);
            if (CMODE == 0) {
#pragma unroll
                for (int k = 0; k < CDIM; ++k)
                    ncol[k] = (ni >= tg.range_start && (uint32_t)k < cnt) ? a.colors[(size_t)nxt.g * a.s_color + ch_off + k] : 0.f;
            }
        }
        __builtin_amdgcn_wave_barrier();

        float4 r0 = s_rec[0], r1 = s_rec[1], r2 = s_rec[2], r3 = s_rec[3];
        for (int j = 0; j < count; ++j) {
            const float4 c0 = r0, c1 = r1, c2 = r2, c3 = r3;
            r0 = s_rec[(j + 1) * REC + 0];
            r1 = s_rec[(j + 1) * REC + 1];
            r2 = s_rec[(j + 1) * REC + 2];
            r3 = s_rec[(j + 1) * REC + 3];
            const int32_t idx = __float_as_int(c2.z);
            const int32_t g = __float_as_int(c2.w);
            const float *cp = a.colors + (size_t)g * a.s_color + ch_off; // wave-uniform
            float col[CR];
            if (CMODE == 0) {
                col[0] = c1.z;
                if (CDIM > 1) col[CDIM > 1 ? 1 : 0] = c1.w;
                if (CDIM > 2) col[CDIM > 2 ? 2 : 0] = c2.x;
                if (CDIM > 3) col[CDIM > 3 ? 3 : 0] = c2.y;
            } else if (CMODE == 1) {
#pragma unroll
                for (int k = 0; k < CR; ++k) col[k] = (uint32_t)k < cnt ? cp[k] : 0.f;
            }
            float S0 = 0.f, Sx = 0.f, Sy = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f, Ax = 0.f, Ay = 0.f;
            float Cs[CR];
#pragma unroll
            for (int k = 0; k < CR; ++k) Cs[k] = 0.f;
            float facs[NQ];
            bool any_valid = false;
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const float dx = c0.x - px[i], dy = c0.y - py[i];
                const float pl = __builtin_fmaf(dx, __builtin_fmaf(c0.w, dy, c0.z * dx), __builtin_fmaf(c1.x * dy, dy, c1.y));
                const float araw = __builtin_amdgcn_exp2f(pl); // = o exp(-sigma)
                const float alpha = fminf(0.999f, araw);
                const bool valid = (idx <= bin_final[i]) && !(pl > c1.y) && (alpha >= ALPHA_MIN);
                any_valid |= valid;
                const float ra = __builtin_amdgcn_rcpf(1.f - alpha);
                const float Tn = T[i] * ra;
                const float facv = valid ? alpha * Tn : 0.f;
                float D = 0.f;
                if (CMODE == 2) {
                    if (valid)
                        for (uint32_t k = 0; k < cnt; ++k) D += cp[k] * ga.v_render_colors[vpix[i] + k * ga.s_vrc_ch];
                } else {
#pragma unroll
                    for (int k = 0; k < CR; ++k) {
                        D += col[k] * vc[i][k];
                        Cs[k] += facv * vc[i][k];
                    }
                }
                facs[i] = facv;
                const float v_alpha = D * Tn + (Tw[i] - Bq[i]) * ra;
                // gradient gate: nothing for conic / xy / opacity when o * vis > 0.999
                const float v_sigma = (valid && araw <= 0.999f) ? -araw * v_alpha : 0.f;
                Bq[i] += facv * D;
                T[i] = valid ? Tn : T[i];
                const float sdx = v_sigma * dx, sdy = v_sigma * dy;
                S0 += v_sigma;
                Sx += sdx;
                Sy += sdy;
                Sxx += sdx * dx;
                Sxy += sdx * dy;
                Syy += sdy * dy;
                if (ABS) {
                    Ax += fabsf(c3.x * sdx + c3.y * sdy);
                    Ay += fabsf(c3.y * sdx + c3.z * sdy);
                }
            }
            if (!__any(any_valid)) continue;
            float *vcol = ga.v_colors + (size_t)g * ga.s_color + ch_off;
            if (CMODE == 2) {
                // any channel count: one reduction + atomic per channel
                for (uint32_t k = 0; k < cnt; ++k) {
                    float c = 0.f;
#pragma unroll
                    for (int i = 0; i < NQ; ++i) c += facs[i] * (inside[i] ? ga.v_render_colors[vpix[i] + k * ga.s_vrc_ch] : 0.f);
                    c = wave_reduce_sum_dpp(c);
                    if (lane == GS_WAVE - 1) unsafeAtomicAdd(vcol + k, c); // (more than 4 channels: no deterministic mode, checked by the caller)
                }
            } else {
#pragma unroll
                for (int k = 0; k < CR; ++k) Cs[k] = wave_reduce_sum_dpp(Cs[k]);
            }
            S0 = wave_reduce_sum_dpp(S0);
            Sx = wave_reduce_sum_dpp(Sx);
            Sy = wave_reduce_sum_dpp(Sy);
            Sxx = wave_reduce_sum_dpp(Sxx);
            Sxy = wave_reduce_sum_dpp(Sxy);
            Syy = wave_reduce_sum_dpp(Syy);
            if (ABS) {
                Ax = wave_reduce_sum_dpp(Ax);
                Ay = wave_reduce_sum_dpp(Ay);
            }
            if (lane == GS_WAVE - 1) {
                const size_t gr = (size_t)g;
                if (CMODE != 2) {
#pragma unroll
                    for (int k = 0; k < CR; ++k)
                        if ((uint32_t)k < cnt) grad_add(ga, vcol + k, gr, 6u + ch_off + (uint32_t)k, Cs[k]);
                }
                grad_add(ga, ga.v_means2d + ga.s_xy * gr, gr, 0u, c3.x * Sx + c3.y * Sy);
                grad_add(ga, ga.v_means2d + ga.s_xy * gr + 1, gr, 1u, c3.y * Sx + c3.z * Sy);
                grad_add(ga, ga.v_conics + ga.s_conic * gr, gr, 2u, 0.5f * Sxx);
                grad_add(ga, ga.v_conics + ga.s_conic * gr + 1, gr, 3u, Sxy);
                grad_add(ga, ga.v_conics + ga.s_conic * gr + 2, gr, 4u, 0.5f * Syy);
                grad_add(ga, ga.v_opacities + ga.s_opac * gr, gr, 5u, -S0 / c3.w);
                if (ABS) {
                    grad_add(ga, ga.v_means2d_abs + ga.s_abs * gr, gr, 10u, Ax);
                    grad_add(ga, ga.v_means2d_abs + ga.s_abs * gr + 1, gr, 11u, Ay);
                }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// more than 4 channels (feature rendering): one quadrant per wave, colours read from global memory
template <int CDIM>
void launch_fwd(const RasterArgs &a, uint32_t cnt, uint32_t off, hipStream_t st) {
    dim3 grid(a.C * a.tile_width * a.tile_height * 4);
    hipLaunchKernelGGL((raster_wave_fwd_kernel<1, CDIM, false, false>), grid, dim3(GS_WAVE), 0, st, a, cnt, off, (float *)nullptr, 0);
}

template <int CDIM, int CMODE>
void launch_bwd(const RasterArgs &a, const RasterGradArgs &ga, uint32_t cnt, uint32_t off, int use_va, hipStream_t st) {
    dim3 grid(a.C * a.tile_width * a.tile_height * 4);
    if (ga.v_means2d_abs != nullptr)
        hipLaunchKernelGGL((raster_wave_bwd_kernel<1, CDIM, CMODE, true>), grid, dim3(GS_WAVE), 0, st, a, ga, cnt, off, use_va);
    else
        hipLaunchKernelGGL((raster_wave_bwd_kernel<1, CDIM, CMODE, false>), grid, dim3(GS_WAVE), 0, st, a, ga, cnt, off, use_va);
}

// ---------------------------------------------------------------------------
// Fast segmented backward for 1..4 channels (the hot case): one wave per (tile, segment) item,
// 4 pixels per lane.  Differences from the generic kernel above:
//   * no compaction: every lane stages its list entry at slot = lane, and FOUR ballots (one per
//     8x8 quadrant, ellipse-extent test against that quadrant) give four 64-bit SGPR masks.  The
//     walk iterates over the set bits of their union (s_ff1) and enters a quadrant's pixel code
//     through a scalar branch -- segments bound the critical path, so throughput is what
//     counts here, and a splat typically touches ~2 of the 4 quadrants;
//   * the 6 + CDIM (+2) per-splat sums are reduced by hand-scheduled interleaved v_add_f32_dpp
//     chains (dpp_reduce.h): 6 instructions per value, no moves, no hazards.
// record: R0 = (mx, my, a', b')  R1 = (c', log2 o, col0, col1)  R2 = (col2, col3, a, b)  R3 = (c, o, g, -)
// ---------------------------------------------------------------------------
#ifndef GS_SEG_WAVES
#define GS_SEG_WAVES 5
#endif
template <int CDIM, bool ABS, bool DET>
__global__ void __launch_bounds__(GS_WAVE, GS_SEG_WAVES) raster_seg_bwd_kernel(RasterArgs a, RasterGradArgs ga, int use_v_alpha, SegArgs sg) {
    constexpr int REC = 4;
    constexpr int ACC = 3; // float4 per accumulator slot: (Sx', Sy', Sxx, Sxy) (Syy, S0, C0, C1) (C2, C3, Ax, Ay)
    __shared__ float4 s_rec[GS_WAVE * REC];
    __shared__ float4 s_acc[GS_WAVE * ACC];
    const uint32_t lane = threadIdx.x;
    const uint32_t lx = lane & 7u, ly = lane >> 3;
    // block index -> item: the classes in order of decreasing cost (the grid is an upper bound on the total)
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

    // issue the first batch's gathers before anything else: their latency overlaps the pixel-state
    // loads below.  The walk starts at the segment's far end; entries behind every pixel's
    // last_ids are dropped by the per-quadrant test (idx <= q_bin_max).
    const int32_t first = tg.range_end - 1;
    auto fetch = [&](int32_t idx, SplatRaw &s, float *col) {
        s.g = 0;
        s.mx = s.my = s.ca = s.cb = s.cc = s.opac = 0.f;
#pragma unroll
        for (int k = 0; k < CDIM; ++k) col[k] = 0.f;
        if (idx >= tg.range_start) fetch_splat<CDIM>(a, a.flatten_ids[idx], s, col);
    };
    SplatRaw nxt;
    float ncol[CDIM];
    fetch(first - (int32_t)lane, nxt, ncol);

    bool inside[4];
    float T[4], Wq[4], vc[4][CDIM]; // Wq = T_final (v_alpha_out - bg . v_out) - B  (the only way Tw and B are used)
    const float px0 = (float)(tg.px0 + lx) + 0.5f, py0 = (float)(tg.py0 + ly) + 0.5f; // quadrant i: + 8 (i&1), + 8 (i>>1)
    int32_t bin_final[4], q_bin_max[4];
    float qx0[4], qx1[4], qy0[4], qy1[4];
    unsigned q_live = 0;
    const float *bg = a.backgrounds ? a.backgrounds + (size_t)tg.cam * a.channels : nullptr;
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
            vc[i][k] = inside[i] ? ga.v_render_colors[pix * ga.s_vrc_pix + k * ga.s_vrc_ch] : 0.f;
            if (bg != nullptr) bg_dot += bg[k] * vc[i][k];
        }
        const float v_a = (inside[i] && use_v_alpha) ? ga.v_render_alphas[pix] : 0.f;
        Wq[i] = T_final * (v_a - bg_dot);
        bin_final[i] = inside[i] ? ga.last_ids[pix] : -1;
        if (from_ckpt && inside[i]) {
            const float *cb = sg.ckpt + (size_t)(seg_k + 1) * (CDIM + 1) * 256 + i * 64 + lane;
            T[i] = cb[0];
            float bsum = 0.f;
#pragma unroll
            for (int k = 0; k < CDIM; ++k) {
                float fin = sg.render_colors[pix * CDIM + k];
                if (bg != nullptr) fin -= T_final * bg[k];
                bsum += vc[i][k] * (fin - cb[(k + 1) * 256]);
            }
            Wq[i] -= bsum;
        }
        q_bin_max[i] = __builtin_amdgcn_readfirstlane(wave_max_i32(bin_final[i])); // make it an SGPR
        if (q_bin_max[i] >= tg.range_start) q_live |= 1u << i;
        // quadrant rectangle (pixel centres), clipped to tile size and image
        const float X0 = (float)(tg.px0 + 8u * (i & 1)) + 0.5f, Y0 = (float)(tg.py0 + 8u * (i >> 1)) + 0.5f;
        auto sgpr = [](float v) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(v))); };
        qx0[i] = sgpr(X0);
        qy0[i] = sgpr(Y0);
        qx1[i] = sgpr(fminf(X0 + 7.f, fminf((float)(tg.px0 + a.tile_size) - 0.5f, (float)a.image_width - 0.5f)));
        qy1[i] = sgpr(fminf(Y0 + 7.f, fminf((float)(tg.py0 + a.tile_size) - 0.5f, (float)a.image_height - 0.5f)));
    }
    if (q_live == 0u) return; // nothing composited in this segment for any pixel
    const int32_t total = first - tg.range_start + 1;
    const int32_t num_batches = (total + GS_WAVE - 1) / GS_WAVE;

    for (int32_t b = 0; b < num_batches; ++b) {
        const int32_t batch_end = first - b * GS_WAVE; // slot t holds list index batch_end - t
        unsigned long long qm[4];
        {
        const SplatRaw s = nxt;
        CullSplat cs;
        const int32_t my_idx = batch_end - (int32_t)lane;
        const bool live = (my_idx >= tg.range_start) && cull_prepare(s, cs);
        // Pixels whose last contributor lies in front of this batch take no part in it (idx <= bin_final fails for every
        // entry): the splats are culled against the bounding rectangle of the pixels that ARE still in play, not against
        // the whole quadrant.  Most pixels saturate a third of the way into their tile's list, so towards the far end of a
        // list a quadrant is often down to a handful of pixels.  Exact: a culled splat has alpha < 1/255 on every pixel
        // that can use it.
        const int32_t batch_lo = max(tg.range_start, batch_end - (GS_WAVE - 1));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const unsigned long long in_play = __ballot(bin_final[i] >= batch_lo);
            qm[i] = 0ull;
            if (((q_live >> i) & 1u) && in_play != 0ull) { // (wave-uniform)
                const LiveRect lr = live_rect(in_play, qx0[i], qy0[i]);
                const bool touch = live && (my_idx <= q_bin_max[i]) && rect_touch(s, cs, lr.x0, lr.x1, lr.y0, lr.y1);
                qm[i] = __ballot(touch);
            }
        }
        {
            float c0 = ncol[0], c1 = 0.f, c2 = 0.f, c3 = 0.f;
            if (CDIM > 1) c1 = ncol[CDIM > 1 ? 1 : 0];
            if (CDIM > 2) c2 = ncol[CDIM > 2 ? 2 : 0];
            if (CDIM > 3) c3 = ncol[CDIM > 3 ? 3 : 0];
            s_rec[lane * REC + 0] = make_float4(s.mx, s.my, -0.5f * LOG2E * s.ca, -LOG2E * s.cb);
            s_rec[lane * REC + 1] = make_float4(-0.5f * LOG2E * s.cc, __log2f(s.opac), c0, c1);
            s_rec[lane * REC + 2] = make_float4(c2, c3, s.ca, s.cb);
            s_rec[lane * REC + 3] = make_float4(s.cc, s.opac, __int_as_float(s.g), 0.f);
        }
        }
        if (b + 1 < num_batches) fetch(first - (b + 1) * GS_WAVE - (int32_t)lane, nxt, ncol);
        __builtin_amdgcn_wave_barrier();

        unsigned long long any = qm[0] | qm[1] | qm[2] | qm[3];
        unsigned long long touched = 0ull; // slots whose sums were stored this batch
        // software pipeline: the next record is read from LDS while the current one is processed
        // (both compositing kernels are bound by the number of instructions a wave issues, scalar ones included -- round 2:
        // one dummy s_add per record costs as much as one dummy v_fma -- so the bookkeeping is kept on as few as possible)
        // The loop is unrolled by two with the roles of the two record register sets swapped (A current / B next, then B
        // current / A next): written as one set copied into the other, the copy was nine v_mov per record.
        struct Rec {
            float4 r0, r1, r2, r3;
        };
        auto load_rec = [&](int slot, Rec &r) {
            r.r0 = s_rec[slot * REC + 0];
            r.r1 = s_rec[slot * REC + 1];
            r.r2 = s_rec[slot * REC + 2];
            r.r3 = ABS ? s_rec[slot * REC + 3] : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        auto next_slot = [&]() { return __builtin_ctzll(any | (1ull << 63)); }; // (an empty set prefetches slot 63: in bounds, never used)
        auto body = [&](const int t, const Rec &rec) {
            const float4 r0 = rec.r0, r1 = rec.r1, r2 = rec.r2, r3 = rec.r3;
            float col[CDIM];
            col[0] = r1.z;
            if (CDIM > 1) col[CDIM > 1 ? 1 : 0] = r1.w;
            if (CDIM > 2) col[CDIM > 2 ? 2 : 0] = r2.x;
            if (CDIM > 3) col[CDIM > 3 ? 3 : 0] = r2.y;
            const int32_t idx = batch_end - t;
            const unsigned long long bit = 1ull << t;
            float S0 = 0.f, Sx = 0.f, Sy = 0.f, Sxx = 0.f, Sxy = 0.f, Syy = 0.f, Ax = 0.f, Ay = 0.f;
            float Cs[CDIM];
#pragma unroll
            for (int k = 0; k < CDIM; ++k) Cs[k] = 0.f;
            float av_sum = 0.f; // > 0 in the lanes with a valid sample in some quadrant (one add per pass; a lane-mask "or" costs four scalar instructions per pass)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!(qm[i] & bit)) continue; // wave-uniform (scalar) branch
                const float dx = r0.x - (px0 + 8.f * (float)(i & 1)), dy = r0.y - (py0 + 8.f * (float)(i >> 1));
                // pl = power + log2(opacity) with the constant riding in the last fma; "sigma < 0" reads pl > log2(opacity)
                // (as in the forward: the two differ only where the sign of a computed sigma is rounding noise)
                const float lo2 = r1.y;
                const float pl = __builtin_fmaf(dx, __builtin_fmaf(r0.w, dy, r0.z * dx), __builtin_fmaf(r1.x * dy, dy, lo2));
                const float araw = __builtin_amdgcn_exp2f(pl); // = o exp(-sigma)
                const float alpha = fminf(0.999f, araw);
                const bool valid = (idx <= bin_final[i]) && !(pl > lo2) && (alpha >= ALPHA_MIN);
                // a rejected record gets alpha = 0: then ra = rcp(1) = 1 exactly, Tn = T and facv = 0, i.e. the
                // transmittance and the colour sums need no select of their own (selects cost 1.5 issue units here)
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
                    Ay += fabsf(r2.w * sdx + r3.x * sdy);
                }
            }
            if (!__any(av_sum > 0.f)) return;
            // 8 values through the permlane butterfly (20 VALU), the rest through plain DPP chains.
            // slot floats: [Sx, Sy | Sxx, Sxy | Syy, S0 | C0, C1 | C2, C3, Ax, Ay]
            float lo, hi;
            const float C1v = CDIM > 1 ? Cs[CDIM > 1 ? 1 : 0] : 0.f;
            float C2v = CDIM > 2 ? Cs[CDIM > 2 ? 2 : 0] : 0.f, C3v = CDIM > 3 ? Cs[CDIM > 3 ? 3 : 0] : 0.f;
            // RGB without absgrad (the hot case): the ninth sum rides in the row reduction of the butterfly and is stored as
            // its four row partials (added up at the flush, once per batch) -- 4 DPP adds instead of a chain of 6 with padding
            constexpr bool ROW3 = !ABS && CDIM == 3;
            if (ROW3) wave_reduce_sum_8_butterfly_rows(Sx, Syy, Sxx, Cs[0], Sy, S0, Sxy, C1v, lo, hi, C2v);
            else wave_reduce_sum_8_butterfly(Sx, Syy, Sxx, Cs[0], Sy, S0, Sxy, C1v, lo, hi);
            if (ROW3) {
            } else if (ABS) {
                if (CDIM > 3) wave_reduce_sum_4(C2v, C3v, Ax, Ay);
                else if (CDIM > 2) wave_reduce_sum_3(C2v, Ax, Ay);
                else wave_reduce_sum_2(Ax, Ay);
            } else {
                if (CDIM > 3) wave_reduce_sum_2(C2v, C3v);
                else if (CDIM > 2) wave_reduce_sum_1(C2v);
            }
            // park the totals in LDS; the atomics are issued once per batch by the lane that staged
            // the splat (full-width atomic instructions instead of 9 one-lane instructions per splat)
            touched |= bit;
            float *acc = reinterpret_cast<float *>(&s_acc[t * ACC]);
            if ((lane & 15u) == 15u) {
                const uint32_t row = lane >> 4; // rows 0..3 hold (v0,v4) (v2,v6) (v1,v5) (v3,v7)
                reinterpret_cast<float2 *>(acc)[row] = make_float2(lo, hi);
                if (ROW3) acc[8 + row] = C2v; // the four row partials of C2 in the slot's third float4
            }
            if (!ROW3 && lane == GS_WAVE - 1) s_acc[t * ACC + 2] = make_float4(C2v, C3v, Ax, Ay);
        };
        {
            Rec ra, rb;
            int ta = next_slot();
            load_rec(ta, ra);
            while (any) {
                any &= ~(1ull << ta); // (the same 1 << t serves the quadrant tests and the `touched` set)
                const int tb = next_slot();
                load_rec(tb, rb);
                body(ta, ra);
                if (!any) break;
                any &= ~(1ull << tb);
                ta = next_slot();
                load_rec(ta, ra);
                body(tb, rb);
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (ga.packed) {
            // Packed gradient rows [n_elems,16]: finalise per slot in LDS, then issue the atomics with
            // lane = (slot, component): the <= 12 components of a splat go out as ONE request to ONE
            // 64-byte line instead of 9-11 requests to 4 arrays (atomics were 27% of this kernel).
            if ((touched >> lane) & 1ull) {
                const float4 a0 = s_acc[lane * ACC + 0], a1 = s_acc[lane * ACC + 1];
                const float4 e2 = s_rec[lane * REC + 2], e3 = s_rec[lane * REC + 3];
                const float e_ca = e2.z, e_cb = e2.w, e_cc = e3.x, e_op = e3.y;
                // row layout: vx vy | ca cb cc | o | c0 c1 c2 c3 | ax ay   (a2 already holds c2 c3 ax ay)
                s_acc[lane * ACC + 0] = make_float4(e_ca * a0.x + e_cb * a0.y, e_cb * a0.x + e_cc * a0.y, 0.5f * a0.z, a0.w);
                s_acc[lane * ACC + 1] = make_float4(0.5f * a1.x, -a1.y / e_op, a1.z, a1.w);
                if (!ABS && CDIM == 3) { // C2 arrives as four row partials
                    const float4 a2 = s_acc[lane * ACC + 2];
                    reinterpret_cast<float *>(&s_acc[lane * ACC + 2])[0] = (a2.x + a2.y) + (a2.z + a2.w);
                }
            }
            __builtin_amdgcn_wave_barrier();
            const uint32_t sub = lane / 12u, comp = lane % 12u; // 5 slots x 12 components per instruction
            const bool comp_on = lane < 60u && (comp < 6u + (uint32_t)CDIM || (ABS && comp >= 10u));
            const float *accf = reinterpret_cast<const float *>(s_acc);
#pragma unroll 1
            for (uint32_t grp = 0; grp < 13u; ++grp) {
                if (((touched >> (grp * 5u)) & 0x1full) == 0ull) continue; // wave-uniform
                const uint32_t slot = grp * 5u + sub;
                if (comp_on && slot < 64u && ((touched >> slot) & 1ull)) {
                    const uint32_t g = (uint32_t)__float_as_int(s_rec[slot * REC + 3].z);
                    grad_add<DET ? 1 : 0>(ga, ga.v_means2d + (size_t)g * 16u + comp, (size_t)g, comp, accf[slot * (ACC * 4) + comp]);
                }
            }
        } else if ((touched >> lane) & 1ull) {
            const float4 a0 = s_acc[lane * ACC + 0], a1 = s_acc[lane * ACC + 1];
            float4 a2 = s_acc[lane * ACC + 2];
            if (!ABS && CDIM == 3) a2.x = (a2.x + a2.y) + (a2.z + a2.w); // C2 arrives as four row partials
            // this lane staged slot `lane`: its raw conic / opacity / id are still in the record
            const float4 e2 = s_rec[lane * REC + 2], e3 = s_rec[lane * REC + 3];
            const size_t g = (size_t)__float_as_int(e3.z);
            const float e_ca = e2.z, e_cb = e2.w, e_cc = e3.x, e_op = e3.y;
            float *vcol = ga.v_colors + g * CDIM;
            grad_add<DET ? 1 : 0>(ga, vcol, g, 6u, a1.z);
            if (CDIM > 1) grad_add<DET ? 1 : 0>(ga, vcol + 1, g, 7u, a1.w);
            if (CDIM > 2) grad_add<DET ? 1 : 0>(ga, vcol + 2, g, 8u, a2.x);
            if (CDIM > 3) grad_add<DET ? 1 : 0>(ga, vcol + 3, g, 9u, a2.y);
            grad_add<DET ? 1 : 0>(ga, ga.v_means2d + ga.s_xy * g, g, 0u, e_ca * a0.x + e_cb * a0.y);
            grad_add<DET ? 1 : 0>(ga, ga.v_means2d + ga.s_xy * g + 1, g, 1u, e_cb * a0.x + e_cc * a0.y);
            grad_add<DET ? 1 : 0>(ga, ga.v_conics + ga.s_conic * g, g, 2u, 0.5f * a0.z);
            grad_add<DET ? 1 : 0>(ga, ga.v_conics + ga.s_conic * g + 1, g, 3u, a0.w);
            grad_add<DET ? 1 : 0>(ga, ga.v_conics + ga.s_conic * g + 2, g, 4u, 0.5f * a1.x);
            grad_add<DET ? 1 : 0>(ga, ga.v_opacities + ga.s_opac * g, g, 5u, -a1.y / e_op);
            if (ABS) {
                grad_add<DET ? 1 : 0>(ga, ga.v_means2d_abs + ga.s_abs * g, g, 10u, a2.z);
                grad_add<DET ? 1 : 0>(ga, ga.v_means2d_abs + ga.s_abs * g + 1, g, 11u, a2.w);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// segmented launch: one wave per (tile, segment) item, 4 pixels per lane
template <int CDIM>
void launch_bwd_seg(const RasterArgs &a, const RasterGradArgs &ga, uint32_t max_items, int use_va, const SegArgs &sg, hipStream_t st) {
    dim3 grid(max_items, 1);
    const bool abs = ga.v_means2d_abs != nullptr, det = ga.det != nullptr;
    if (abs && det) hipLaunchKernelGGL((raster_seg_bwd_kernel<CDIM, true, true>), grid, dim3(GS_WAVE), 0, st, a, ga, use_va, sg);
    else if (abs) hipLaunchKernelGGL((raster_seg_bwd_kernel<CDIM, true, false>), grid, dim3(GS_WAVE), 0, st, a, ga, use_va, sg);
    else if (det) hipLaunchKernelGGL((raster_seg_bwd_kernel<CDIM, false, true>), grid, dim3(GS_WAVE), 0, st, a, ga, use_va, sg);
    else hipLaunchKernelGGL((raster_seg_bwd_kernel<CDIM, false, false>), grid, dim3(GS_WAVE), 0, st, a, ga, use_va, sg);
}

// Work list of the segmented backward, ORDERED BY COST, longest first.  Why: a (tile, segment) item lasts 20 ... 120 us
// (p10 ... max) and the hardware starts workgroups in index order, so with a list in arrival order the kernel ended with a
// ~95 us drain of long items at low occupancy; longest-first (LPT) leaves the short items for the end (measured: 365 -> 329 us).
// cost = what the forward counted for the item (records evaluated by the four quadrant waves).  Items are bucketed into
// COST_CLASSES classes, each with its own region of the item array and a counter (zeroed by the forward kernel); the
// backward maps its block index through the class counts, most expensive class first.  Built right after the forward
// kernel, inside gs_rasterize_fwd: the backward itself is ONE launch, and a repeated backward (retain_graph) reuses it.
//   unit u < n_tiles_all           : the first segment of tile u
//   unit u = n_tiles_all + k       : segment k of the tile that owns list boundary k * seg (body_tile[k], verified)
__global__ void __launch_bounds__(GS_BLOCK) seg_items_build_kernel(uint32_t n_tiles_all, uint32_t n_isects, const int32_t *__restrict__ offsets,
                                                                   const uint8_t *__restrict__ masks, int32_t seg, uint32_t n_bounds,
                                                                   const uint32_t *__restrict__ cost_head, const uint32_t *__restrict__ cost_body,
                                                                   const uint32_t *__restrict__ body_tile, uint32_t *__restrict__ class_count,
                                                                   uint2 *__restrict__ items, uint32_t max_items) {
    __shared__ uint32_t s_cnt[COST_CLASSES], s_base[COST_CLASSES];
    if (threadIdx.x < COST_CLASSES) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    const uint32_t u = blockIdx.x * GS_BLOCK + threadIdx.x;
    uint32_t tile = 0, cls = 0, local = 0;
    int32_t k = 0;
    bool valid = false;
    auto range = [&](uint32_t t, int32_t &rs, int32_t &re) {
        rs = offsets[t];
        re = (t + 1 == n_tiles_all) ? (int32_t)n_isects : offsets[t + 1];
    };
    if (u < n_tiles_all) {
        int32_t rs, re;
        range(u, rs, re);
        valid = re > rs && (masks == nullptr || masks[u]);
        tile = u;
        k = rs / seg;
        if (valid) {
            const uint4 c = reinterpret_cast<const uint4 *>(cost_head)[u];
            cls = cost_class(c.x + c.y + c.z + c.w, seg);
        }
    } else if (u - n_tiles_all < n_bounds) {
        k = (int32_t)(u - n_tiles_all);
        const uint32_t t = body_tile[k]; // stale or never written unless boundary k * seg lies strictly inside a live tile
        if (k > 0 && t < n_tiles_all) {
            int32_t rs, re;
            range(t, rs, re);
            valid = (int64_t)rs < (int64_t)k * seg && (int64_t)k * seg < (int64_t)re && (masks == nullptr || masks[t]);
            tile = t;
            if (valid) {
                const uint4 c = reinterpret_cast<const uint4 *>(cost_body)[k];
                cls = cost_class(c.x + c.y + c.z + c.w, seg);
            }
        }
    }
    if (valid) local = atomicAdd(&s_cnt[cls], 1u);
    __syncthreads();
    if (threadIdx.x < COST_CLASSES) {
        const uint32_t c = s_cnt[threadIdx.x];
        s_base[threadIdx.x] = c ? atomicAdd(&class_count[threadIdx.x], c) : 0u;
    }
    __syncthreads();
    if (valid) items[(size_t)cls * max_items + s_base[cls] + local] = make_uint2(tile, (uint32_t)k);
}


// Deterministic mode, second pass: fixed-point sums -> the float gradient outputs (every row written: no zero-fill needed)
__global__ void __launch_bounds__(GS_BLOCK) raster_det_finalize_kernel(uint32_t n_elems, uint32_t channels, const long long *__restrict__ det,
                                                                       RasterGradArgs ga) {
    const uint32_t r = blockIdx.x * GS_BLOCK + threadIdx.x;
    if (r >= n_elems) return;
    const long long *d = det + (size_t)r * 24u;
    auto f = [&](uint32_t c) { return (float)((double)d[c] * GS_DET_INV_LO + (double)d[12u + c] * GS_DET_INV_HI); };
    ga.v_means2d[ga.s_xy * (size_t)r] = f(0);
    ga.v_means2d[ga.s_xy * (size_t)r + 1] = f(1);
    ga.v_conics[ga.s_conic * (size_t)r] = f(2);
    ga.v_conics[ga.s_conic * (size_t)r + 1] = f(3);
    ga.v_conics[ga.s_conic * (size_t)r + 2] = f(4);
    ga.v_opacities[ga.s_opac * (size_t)r] = f(5);
    for (uint32_t k = 0; k < channels; ++k) ga.v_colors[ga.s_color * (size_t)r + k] = f(6u + k);
    if (ga.v_means2d_abs != nullptr) {
        ga.v_means2d_abs[ga.s_abs * (size_t)r] = f(10);
        ga.v_means2d_abs[ga.s_abs * (size_t)r + 1] = f(11);
    }
}

} // namespace

// ---------------------------------------------------------------------------
// host side
// scratch layout (the SAME buffer and the SAME plan must be handed to gs_rasterize_fwd and to the matching
// gs_rasterize_bwd; the buffer's contents must be preserved in between):
//   [0, 256)                      item counters of the 32 cost classes (uint32) + padding
//   [256, 256 + items)            (tile, k) work items of the segmented backward (uint2), one region per cost class
//   [.., .. + cost)               cost of every work item as counted by the forward ([tile][4] + [boundary][4] uint32)
//                                 and the owner tile of every list boundary
//   [.., .. + ckpt)               forward checkpoints, (n_isects / seg + 2) x (channels + 1) x 256 floats
// ---------------------------------------------------------------------------
namespace {

// Tuning defaults = the measured optima on MI355X (profiles/round1_notes.md, round2_notes.md).  They travel in the
// caller's gs_raster_plan (gs_rasterize_plan): the library itself holds no tuning state.
//   seg        segment length of the depth-segmented backward in list entries (multiple of 64; 0: no segments,
//              the generic one-quadrant-per-wave backward runs instead).  256 since the work list is ordered
//              longest-first (round 2: 128 / 192 / 256 / 320 / 384 / 512 -> 0.338 / 0.317 / 0.305 / 0.312 / 0.324 /
//              0.331 ms backward, and half the checkpoint planes in the forward); with the list in arrival order
//              128 was best (round 1: 512 -> 1.14 ms, 256 -> 0.97 ms, 128 -> 0.89 ms).
//   solo_min   list length from which a tile's four forward waves stop cooperating (0: never). 2048.
//   xcd_fwd / xcd_bwd   work items per XCD group (xcd_remap).  16 tiles / 16 segment items.
constexpr uint32_t PLAN_MAGIC = 0x47535033u; // "GSP3"

struct ScratchLayout {
    size_t off_items, off_cost_head, off_cost_body, off_body_tile, off_ckpt, off_order, total;
    uint32_t max_items, n_bounds;
};

// channel counts the tile forward / segmented backward cover: 1..4 in one launch, 5..16 in one launch of the wide instances,
// 17..32 as two launches over halves (rasterize_wide.hip)
constexpr uint32_t FAST_MAX_CHANNELS = 32;

ScratchLayout scratch_layout(uint32_t n_tiles_all, uint32_t n_isects, uint32_t channels, int32_t seg) {
    ScratchLayout L;
    auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
    size_t o = 256;
    L.n_bounds = (seg > 0 ? n_isects / (uint32_t)seg : 0) + 2; // list boundaries k * seg, k < n_bounds
    L.max_items = n_tiles_all + L.n_bounds;
    L.off_items = o;
    if (seg > 0 && channels <= FAST_MAX_CHANNELS) o += up((size_t)COST_CLASSES * L.max_items * sizeof(uint2));
    L.off_cost_head = o; // per (tile, quadrant wave): cost of the tile's first backward segment
    o += up((size_t)n_tiles_all * 4 * sizeof(uint32_t));
    L.off_cost_body = o; // per (segment boundary, quadrant wave): cost of the later segments
    if (seg > 0 && channels <= FAST_MAX_CHANNELS) o += up((size_t)L.n_bounds * 4 * sizeof(uint32_t));
    L.off_body_tile = o; // per segment boundary: the tile that owns it
    if (seg > 0 && channels <= FAST_MAX_CHANNELS) o += up((size_t)L.n_bounds * sizeof(uint32_t));
    L.off_ckpt = o;
    if (seg > 0 && channels <= FAST_MAX_CHANNELS) o += up(((size_t)n_isects / seg + 2) * (channels + 1) * 256 * sizeof(float));
    L.off_order = o; // the forward's tile order, heaviest lists first (tile_order_kernel)
    if (channels <= FAST_MAX_CHANNELS) o += up((size_t)n_tiles_all * sizeof(uint32_t));
    L.total = o;
    return L;
}

ScratchLayout scratch_layout(const gs_raster_plan &p) { return scratch_layout(p.n_tiles_all, p.n_isects, p.channels, p.seg); }

} // namespace

int32_t raster_make_plan(uint32_t n_tiles_all, uint32_t n_isects, uint32_t channels, const int32_t *tuning, gs_raster_plan *plan) {
    memset(plan, 0, sizeof(*plan));
    // forward tiles longest list first: worth its one-workgroup ordering launch (10 us at 8 K tiles, 60 us at 65 K) while a batch has few
    // enough tiles for the tail to matter -- measured on BASELINE config 2's scene: +0.5 % at 1 camera, neutral at 2-3, -1.2 % at 4 and
    // -2.3 % at 8 cameras (32 K / 65 K tiles), so: up to three 1080p cameras' worth of tiles
    int32_t seg = 256, solo = 2048, xf = 16, xb = 16, order = n_tiles_all <= 24576u;
    if (tuning != nullptr) {
        if (tuning[0] >= 0) seg = ((tuning[0] + 63) / 64) * 64;
        if (tuning[1] >= 0) solo = tuning[1];
        if (tuning[2] >= 0) xf = tuning[2];
        if (tuning[3] >= 0) xb = tuning[3];
        if (tuning[4] >= 0) order = tuning[4] != 0;
    }
    // the segment length is doubled until the checkpoint array stays below 65536 boundaries (256 MB for RGB; the same
    // byte bound for more planes: 5..32 channels)
    while (seg > 0 && (uint64_t)n_isects / (uint32_t)seg * (channels > 4u ? channels + 1u : 4u) > 65536u * 4u) seg *= 2;
    plan->magic = PLAN_MAGIC;
    plan->n_tiles_all = n_tiles_all;
    plan->n_isects = n_isects;
    plan->channels = channels;
    plan->seg = seg;
    plan->solo_min = solo;
    plan->xcd_fwd = (uint32_t)xf;
    plan->xcd_bwd = (uint32_t)xb;
    plan->reserved[0] = (uint32_t)order; // forward: tiles with the longest lists first
    plan->scratch_bytes = scratch_layout(*plan).total;
    return 0;
}

bool raster_plan_ok(const gs_raster_plan *plan, uint32_t n_tiles_all, uint32_t n_isects, uint32_t channels) {
    return plan != nullptr && plan->magic == PLAN_MAGIC && plan->n_tiles_all == n_tiles_all && plan->n_isects == n_isects &&
           plan->channels == channels && plan->seg >= 0 && plan->seg % 64 == 0 &&
           plan->scratch_bytes == scratch_layout(*plan).total;
}

int32_t raster_wave_fwd(const RasterArgs &a_in, const gs_raster_plan *plan, void *scratch, void *zero_fill, size_t zero_fill_bytes,
                        hipStream_t st) {
    RasterArgs a = a_in;
    const uint32_t n_tiles_all = a.C * a.tile_width * a.tile_height;
    gs_raster_plan dflt; // no plan: default launch geometry, no checkpoints
    if (plan == nullptr) {
        raster_make_plan(n_tiles_all, a.n_isects, a.channels, nullptr, &dflt);
        scratch = nullptr;
    }
    const gs_raster_plan &P = plan ? *plan : dflt;
    const int32_t seg = P.seg;
    const bool ckpt_on = a.channels <= FAST_MAX_CHANNELS && seg > 0 && scratch != nullptr;
    // the side job: spread over the tile workgroups when each gets at most 256 KB of it, a plain fill otherwise
    ZeroFill zf = {nullptr, 0, 0u};
    if (zero_fill != nullptr && zero_fill_bytes > 0) {
        const size_t n16 = zero_fill_bytes / 16;
        const size_t per = (n16 + n_tiles_all - 1) / (n_tiles_all ? n_tiles_all : 1);
        const bool in_kernel = ckpt_on && n_tiles_all > 0 && per <= 16384;
        if (in_kernel) zf = {(float4 *)zero_fill, n16, (uint32_t)per};
        else if (hipMemsetAsync(zero_fill, 0, zero_fill_bytes, st) != hipSuccess) { gs_set_error("rasterize: zero fill failed"); return 1; }
    }
    if (a.channels <= 4) {
        // one 256-thread workgroup per tile; checkpoints for the segmented backward when the caller handed over scratch
        const ScratchLayout L = scratch_layout(P);
        float *ckpt = ckpt_on ? (float *)((char *)scratch + L.off_ckpt) : nullptr;
        a.xcd_group = P.xcd_fwd;
        a.tile_order = nullptr;
        if (scratch != nullptr && P.reserved[0] != 0u && n_tiles_all > 0) {
            uint32_t *order = (uint32_t *)((char *)scratch + L.off_order);
            hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(ORDER_THREADS), 0, st, n_tiles_all, a.tile_offsets, a.n_isects, order);
            a.tile_order = order;
        }
        const int32_t solo = P.solo_min;
        uint32_t *ch = ckpt ? (uint32_t *)((char *)scratch + L.off_cost_head) : nullptr;
        uint32_t *cb = ckpt ? (uint32_t *)((char *)scratch + L.off_cost_body) : nullptr;
        uint32_t *bt = ckpt ? (uint32_t *)((char *)scratch + L.off_body_tile) : nullptr;
        uint32_t *cc = ckpt ? (uint32_t *)scratch : nullptr;
        switch (a.channels) {
            case 1: launch_tile_fwd<1>(a, ckpt, seg, solo, ch, cb, bt, cc, zf, st); break;
            case 2: launch_tile_fwd<2>(a, ckpt, seg, solo, ch, cb, bt, cc, zf, st); break;
            case 3: launch_tile_fwd<3>(a, ckpt, seg, solo, ch, cb, bt, cc, zf, st); break;
            default: launch_tile_fwd<4>(a, ckpt, seg, solo, ch, cb, bt, cc, zf, st); break;
        }
        if (ckpt != nullptr) // the backward's work list, ordered by the costs just counted
            hipLaunchKernelGGL(seg_items_build_kernel, dim3(gs_div_up(L.max_items, GS_BLOCK)), dim3(GS_BLOCK), 0, st, n_tiles_all, a.n_isects,
                               a.tile_offsets, a.masks, seg, L.n_bounds, ch, cb, bt, cc, (uint2 *)((char *)scratch + L.off_items), L.max_items);
        return 0;
    }
    if (a.channels <= FAST_MAX_CHANNELS && plan != nullptr) {
        // 5..32 channels (round 5): the tile kernel with the colours in the LDS record; 17..32 as two launches over halves
        const ScratchLayout L = scratch_layout(P);
        float *ckpt = ckpt_on ? (float *)((char *)scratch + L.off_ckpt) : nullptr;
        a.xcd_group = P.xcd_fwd;
        a.tile_order = nullptr;
        if (scratch != nullptr && P.reserved[0] != 0u && n_tiles_all > 0) {
            uint32_t *order = (uint32_t *)((char *)scratch + L.off_order);
            hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(ORDER_THREADS), 0, st, n_tiles_all, a.tile_offsets, a.n_isects, order);
            a.tile_order = order;
        }
        uint32_t *ch = ckpt ? (uint32_t *)((char *)scratch + L.off_cost_head) : nullptr;
        uint32_t *cb = ckpt ? (uint32_t *)((char *)scratch + L.off_cost_body) : nullptr;
        uint32_t *bt = ckpt ? (uint32_t *)((char *)scratch + L.off_body_tile) : nullptr;
        uint32_t *cc = ckpt ? (uint32_t *)scratch : nullptr;
        const uint32_t n_chunks = a.channels > 16u ? 2u : 1u, first = (a.channels + n_chunks - 1u) / n_chunks;
        for (uint32_t off = 0; off < a.channels; off += first) {
            const uint32_t cnt = min(first, a.channels - off);
            const ZeroFill z = off == 0u ? zf : ZeroFill{nullptr, 0, 0u};
            if (cnt <= 8u) launch_tile_fwd<8>(a, ckpt, seg, P.solo_min, ch, cb, bt, cc, z, st, off, cnt);
            else if (cnt == 9u) launch_tile_fwd<9>(a, ckpt, seg, P.solo_min, ch, cb, bt, cc, z, st, off, cnt);
            else if (cnt <= 12u) launch_tile_fwd<12>(a, ckpt, seg, P.solo_min, ch, cb, bt, cc, z, st, off, cnt);
            else launch_tile_fwd<16>(a, ckpt, seg, P.solo_min, ch, cb, bt, cc, z, st, off, cnt);
        }
        if (ckpt != nullptr)
            hipLaunchKernelGGL(seg_items_build_kernel, dim3(gs_div_up(L.max_items, GS_BLOCK)), dim3(GS_BLOCK), 0, st, n_tiles_all, a.n_isects,
                               a.tile_offsets, a.masks, seg, L.n_bounds, ch, cb, bt, cc, (uint2 *)((char *)scratch + L.off_items), L.max_items);
        return 0;
    }
    // (no plan, or more than 32 channels) one quadrant per wave, exact chunks of 32 channels
    a.xcd_group = P.xcd_fwd * 4u;
    for (uint32_t off = 0; off < a.channels; off += 32) {
        uint32_t cnt = min(32u, a.channels - off);
        if (cnt <= 8) launch_fwd<8>(a, cnt, off, st);
        else if (cnt <= 16) launch_fwd<16>(a, cnt, off, st);
        else launch_fwd<32>(a, cnt, off, st);
    }
    return 0;
}

int32_t raster_wave_bwd(const RasterArgs &a_in, const RasterGradArgs &ga, const float *render_colors, const gs_raster_plan *plan,
                        void *scratch, hipStream_t st) {
    RasterArgs a = a_in;
    const uint32_t n_tiles_all = a.C * a.tile_width * a.tile_height;
    gs_raster_plan dflt;
    if (plan == nullptr) {
        raster_make_plan(n_tiles_all, a.n_isects, a.channels, nullptr, &dflt);
        scratch = nullptr;
    }
    const gs_raster_plan &P = plan ? *plan : dflt;
    a.xcd_group = P.xcd_bwd;
    const int use_va = ga.v_render_alphas != nullptr;
    const uint32_t c = a.channels;
    const int32_t seg = P.seg;
    // Depth-segmented backward: needs the forward's checkpoints (same plan, same scratch) and the render.
    if (seg > 0 && c <= 4 && scratch != nullptr && render_colors != nullptr) {
        const ScratchLayout L = scratch_layout(P);
        // the work list was built by gs_rasterize_fwd (seg_items_build_kernel): ONE launch here
        SegArgs sg = {(const uint2 *)((char *)scratch + L.off_items), (const uint32_t *)scratch, L.max_items,
                      (const float *)((char *)scratch + L.off_ckpt), render_colors, seg};
        switch (c) {
            case 1: launch_bwd_seg<1>(a, ga, L.max_items, use_va, sg, st); break;
            case 2: launch_bwd_seg<2>(a, ga, L.max_items, use_va, sg, st); break;
            case 3: launch_bwd_seg<3>(a, ga, L.max_items, use_va, sg, st); break;
            default: launch_bwd_seg<4>(a, ga, L.max_items, use_va, sg, st); break;
        }
        if (ga.det != nullptr)
            hipLaunchKernelGGL(raster_det_finalize_kernel, dim3(gs_div_up(a.n_elems, GS_BLOCK)), dim3(GS_BLOCK), 0, st, a.n_elems, c, ga.det, ga);
        return 0;
    }
    // 5..32 channels: the wide segmented kernel (rasterize_wide.hip); absgrad is not linear in the image gradient, so with it
    // only what fits ONE launch (<= 16 channels) goes this way
    if (seg > 0 && c > 4 && c <= FAST_MAX_CHANNELS && scratch != nullptr && render_colors != nullptr && ga.det == nullptr &&
        (c <= 16 || ga.v_means2d_abs == nullptr)) {
        const ScratchLayout L = scratch_layout(P);
        const uint32_t n_chunks = c > 16u ? 2u : 1u, first = (c + n_chunks - 1u) / n_chunks;
        for (uint32_t off = 0; off < c; off += first)
            raster_seg_bwd_wide(a, ga, L.max_items, (use_va && off == 0u) ? 1 : 0, (const char *)scratch + L.off_items, (const uint32_t *)scratch,
                                (const float *)((char *)scratch + L.off_ckpt), render_colors, seg, off, min(first, c - off), st);
        return 0;
    }
    // no checkpoints (forward ran without scratch, or segments are switched off) or more than 32 channels:
    // one quadrant per wave walking the whole list back to front
    a.xcd_group = P.xcd_bwd * 4u;
    if (c <= 4) {
        switch (c) {
            case 1: launch_bwd<1, 0>(a, ga, 1, 0, use_va, st); break;
            case 2: launch_bwd<2, 0>(a, ga, 2, 0, use_va, st); break;
            case 3: launch_bwd<3, 0>(a, ga, 3, 0, use_va, st); break;
            default: launch_bwd<4, 0>(a, ga, 4, 0, use_va, st); break;
        }
    } else if (c <= 8) {
        launch_bwd<8, 1>(a, ga, c, 0, use_va, st);
    } else if (c <= 16) {
        launch_bwd<16, 1>(a, ga, c, 0, use_va, st);
    } else if (c <= 32) {
        launch_bwd<32, 1>(a, ga, c, 0, use_va, st);
    } else {
        launch_bwd<1, 2>(a, ga, c, 0, use_va, st);
    }
    if (ga.det != nullptr)
        hipLaunchKernelGGL(raster_det_finalize_kernel, dim3(gs_div_up(a.n_elems, GS_BLOCK)), dim3(GS_BLOCK), 0, st, a.n_elems, c, ga.det, ga);
    return 0;
}
