// rasterize_ref.hip -- R5 baseline: one thread per pixel, one workgroup per tile.
//
// This is the straightforward formulation of per-tile front-to-back alpha compositing
// (same parallelisation as gsplat/cuda/csrc/rasterize_to_pixels_{fwd,bwd}.cu, re-thought
// for wave64: 64-pixel DPP reductions, one atomic group per wave).  It is kept as the
// always-correct reference implementation on the GPU (selected with
// GS_RASTER_IMPL=ref) and as the A/B partner of the wave-per-tile kernels in
// rasterize.hip, which are the default.
#include "gs_common.h"
#include "rasterize_common.h"

namespace {

template <int CDIM>
__global__ void __launch_bounds__(GS_BLOCK) raster_ref_fwd_kernel(RasterArgs a, uint32_t cnt, uint32_t ch_off) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t block_size = blockDim.x;
    int32_t *s_id = (int32_t *)smem;
    float4 *s_geo = (float4 *)(smem + 16 * ((block_size * 4 + 15) / 16)); // x, y, opac, conic.a
    float2 *s_con = (float2 *)(s_geo + block_size);                       // conic.b, conic.c

    const uint32_t cam = blockIdx.z;
    const uint32_t tile_id = blockIdx.y * a.tile_width + blockIdx.x;
    const uint32_t tr = threadIdx.x;
    const uint32_t ty = tr / a.tile_size, tx = tr % a.tile_size;
    const uint32_t i = blockIdx.y * a.tile_size + ty;
    const uint32_t j = blockIdx.x * a.tile_size + tx;
    const bool inside = (tr < a.tile_size * a.tile_size) && i < a.image_height && j < a.image_width;
    const size_t pix = ((size_t)cam * a.image_height + i) * a.image_width + j;
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;

    const float *bg = a.backgrounds ? a.backgrounds + (size_t)cam * a.channels + ch_off : nullptr;
    if (a.masks != nullptr && !a.masks[(size_t)cam * a.tile_width * a.tile_height + tile_id]) {
        // masked tile: background only; alphas / last_ids untouched (reference behaviour)
        if (inside)
            for (uint32_t k = 0; k < cnt; ++k) a.render_colors[pix * a.channels + ch_off + k] = bg ? bg[k] : 0.f;
        return;
    }

    const size_t tile_lin = (size_t)cam * a.tile_width * a.tile_height + tile_id;
    const int32_t range_start = a.tile_offsets[tile_lin];
    const int32_t range_end = (tile_lin + 1 == (size_t)a.C * a.tile_width * a.tile_height)
                                  ? (int32_t)a.n_isects
                                  : a.tile_offsets[tile_lin + 1];
    const uint32_t num_batches = (range_end - range_start + block_size - 1) / block_size;

    float T = 1.f;
    uint32_t cur_idx = 0;
    bool done = !inside;
    float out[CDIM];
#pragma unroll
    for (int k = 0; k < CDIM; ++k) out[k] = 0.f;

    for (uint32_t b = 0; b < num_batches; ++b) {
        if (__syncthreads_count(done) >= (int)block_size) break;
        const uint32_t batch_start = range_start + block_size * b;
        const uint32_t idx = batch_start + tr;
        if (idx < (uint32_t)range_end) {
            int32_t g = a.flatten_ids[idx];
            s_id[tr] = g;
            float2 xy = reinterpret_cast<const float2 *>(a.means2d)[g];
            const float *cn = a.conics + 3 * (size_t)g;
            s_geo[tr] = make_float4(xy.x, xy.y, a.opacities[g], cn[0]);
            s_con[tr] = make_float2(cn[1], cn[2]);
        }
        __syncthreads();
        const uint32_t batch_size = min(block_size, (uint32_t)range_end - batch_start);
        for (uint32_t t = 0; t < batch_size && !done; ++t) {
            const float4 geo = s_geo[t];
            const float2 con = s_con[t];
            const float dx = geo.x - px, dy = geo.y - py;
            const float sigma = 0.5f * (geo.w * dx * dx + con.y * dy * dy) + con.x * dx * dy;
            const float alpha = fminf(0.999f, geo.z * __expf(-sigma));
            if (sigma < 0.f || alpha < 1.f / 255.f) continue;
            const float next_T = T * (1.f - alpha);
            if (next_T <= 1e-4f) {
                done = true;
                break;
            }
            const float vis = alpha * T;
            const float *c = a.colors + (size_t)s_id[t] * a.channels + ch_off;
#pragma unroll
            for (int k = 0; k < CDIM; ++k)
                if ((uint32_t)k < cnt) out[k] += c[k] * vis;
            cur_idx = batch_start + t;
            T = next_T;
        }
    }
    if (inside) {
        a.render_alphas[pix] = 1.f - T;
#pragma unroll
        for (int k = 0; k < CDIM; ++k)
            if ((uint32_t)k < cnt) a.render_colors[pix * a.channels + ch_off + k] = bg ? out[k] + T * bg[k] : out[k];
        a.last_ids[pix] = (int32_t)cur_idx;
    }
}

template <int CDIM>
__global__ void __launch_bounds__(GS_BLOCK) raster_ref_bwd_kernel(RasterArgs a, RasterGradArgs ga, uint32_t cnt, uint32_t ch_off, int use_v_alpha) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t block_size = blockDim.x;
    int32_t *s_id = (int32_t *)smem;
    float4 *s_geo = (float4 *)(smem + 16 * ((block_size * 4 + 15) / 16));
    float2 *s_con = (float2 *)(s_geo + block_size);
    float *s_rgb = (float *)(s_con + block_size); // [block_size][CDIM]

    const uint32_t cam = blockIdx.z;
    const uint32_t tile_id = blockIdx.y * a.tile_width + blockIdx.x;
    if (a.masks != nullptr && !a.masks[(size_t)cam * a.tile_width * a.tile_height + tile_id]) return;
    const uint32_t tr = threadIdx.x;
    const uint32_t ty = tr / a.tile_size, tx = tr % a.tile_size;
    const uint32_t i = blockIdx.y * a.tile_size + ty;
    const uint32_t j = blockIdx.x * a.tile_size + tx;
    const bool inside = (tr < a.tile_size * a.tile_size) && i < a.image_height && j < a.image_width;
    const size_t pix = inside ? ((size_t)cam * a.image_height + i) * a.image_width + j : 0;
    const float px = (float)j + 0.5f, py = (float)i + 0.5f;

    const size_t tile_lin = (size_t)cam * a.tile_width * a.tile_height + tile_id;
    const int32_t range_start = a.tile_offsets[tile_lin];
    const int32_t range_end = (tile_lin + 1 == (size_t)a.C * a.tile_width * a.tile_height)
                                  ? (int32_t)a.n_isects
                                  : a.tile_offsets[tile_lin + 1];
    const uint32_t num_batches = (range_end - range_start + block_size - 1) / block_size;

    const float T_final = inside ? 1.f - ga.render_alphas[pix] : 1.f;
    float T = T_final;
    float buffer[CDIM], v_c[CDIM];
#pragma unroll
    for (int k = 0; k < CDIM; ++k) {
        buffer[k] = 0.f;
        v_c[k] = (inside && (uint32_t)k < cnt) ? ga.v_render_colors[(int64_t)pix * ga.s_vrc_pix + (int64_t)(ch_off + k) * ga.s_vrc_ch] : 0.f;
    }
    const float v_a = (inside && use_v_alpha) ? ga.v_render_alphas[pix] : 0.f;
    const int32_t bin_final = inside ? ga.last_ids[pix] : 0;
    float bg_dot = 0.f;
    if (a.backgrounds != nullptr) {
        const float *bg = a.backgrounds + (size_t)cam * a.channels + ch_off;
#pragma unroll
        for (int k = 0; k < CDIM; ++k)
            if ((uint32_t)k < cnt) bg_dot += bg[k] * v_c[k];
    }
    const int32_t wave_bin_final = wave_max_i32(bin_final);
    const uint32_t lane = tr % GS_WAVE;

    for (uint32_t b = 0; b < num_batches; ++b) {
        __syncthreads();
        const int32_t batch_end = range_end - 1 - (int32_t)(block_size * b);
        const int32_t batch_size = min((int32_t)block_size, batch_end + 1 - range_start);
        const int32_t idx = batch_end - (int32_t)tr;
        if (idx >= range_start) {
            int32_t g = a.flatten_ids[idx];
            s_id[tr] = g;
            float2 xy = reinterpret_cast<const float2 *>(a.means2d)[g];
            const float *cn = a.conics + 3 * (size_t)g;
            s_geo[tr] = make_float4(xy.x, xy.y, a.opacities[g], cn[0]);
            s_con[tr] = make_float2(cn[1], cn[2]);
            const float *c = a.colors + (size_t)g * a.channels + ch_off;
#pragma unroll
            for (int k = 0; k < CDIM; ++k) s_rgb[tr * CDIM + k] = (uint32_t)k < cnt ? c[k] : 0.f;
        }
        __syncthreads();
        for (int32_t t = max(0, batch_end - wave_bin_final); t < batch_size; ++t) {
            bool valid = inside && (batch_end - t <= bin_final);
            float alpha = 0.f, opac = 0.f, vis = 0.f, dx = 0.f, dy = 0.f;
            float ca = 0.f, cb = 0.f, cc = 0.f;
            if (valid) {
                const float4 geo = s_geo[t];
                const float2 con = s_con[t];
                opac = geo.z;
                ca = geo.w; cb = con.x; cc = con.y;
                dx = geo.x - px; dy = geo.y - py;
                const float sigma = 0.5f * (ca * dx * dx + cc * dy * dy) + cb * dx * dy;
                vis = __expf(-sigma);
                alpha = fminf(0.999f, opac * vis);
                if (sigma < 0.f || alpha < 1.f / 255.f) valid = false;
            }
            if (!__any(valid)) continue;
            float v_rgb[CDIM];
#pragma unroll
            for (int k = 0; k < CDIM; ++k) v_rgb[k] = 0.f;
            float v_ca = 0.f, v_cb = 0.f, v_cc = 0.f, v_x = 0.f, v_y = 0.f, v_ax = 0.f, v_ay = 0.f, v_o = 0.f;
            if (valid) {
                const float ra = 1.f / (1.f - alpha);
                T *= ra;
                const float fac = alpha * T;
                float v_alpha = 0.f;
#pragma unroll
                for (int k = 0; k < CDIM; ++k) {
                    v_rgb[k] = fac * v_c[k];
                    v_alpha += (s_rgb[t * CDIM + k] * T - buffer[k] * ra) * v_c[k];
                }
                v_alpha += T_final * ra * v_a;
                v_alpha += -T_final * ra * bg_dot;
                if (opac * vis <= 0.999f) {
                    const float v_sigma = -opac * vis * v_alpha;
                    v_ca = 0.5f * v_sigma * dx * dx;
                    v_cb = v_sigma * dx * dy;
                    v_cc = 0.5f * v_sigma * dy * dy;
                    v_x = v_sigma * (ca * dx + cb * dy);
                    v_y = v_sigma * (cb * dx + cc * dy);
                    v_ax = fabsf(v_x);
                    v_ay = fabsf(v_y);
                    v_o = vis * v_alpha;
                }
#pragma unroll
                for (int k = 0; k < CDIM; ++k) buffer[k] += s_rgb[t * CDIM + k] * fac;
            }
            // 64-lane reductions, then lane 63 issues the atomics
#pragma unroll
            for (int k = 0; k < CDIM; ++k) v_rgb[k] = wave_reduce_sum_dpp(v_rgb[k]);
            v_ca = wave_reduce_sum_dpp(v_ca);
            v_cb = wave_reduce_sum_dpp(v_cb);
            v_cc = wave_reduce_sum_dpp(v_cc);
            v_x = wave_reduce_sum_dpp(v_x);
            v_y = wave_reduce_sum_dpp(v_y);
            v_o = wave_reduce_sum_dpp(v_o);
            if (ga.v_means2d_abs != nullptr) {
                v_ax = wave_reduce_sum_dpp(v_ax);
                v_ay = wave_reduce_sum_dpp(v_ay);
            }
            if (lane == GS_WAVE - 1) {
                const size_t g = (size_t)s_id[t];
                float *vc = ga.v_colors + g * ga.s_color + ch_off;
#pragma unroll
                for (int k = 0; k < CDIM; ++k)
                    if ((uint32_t)k < cnt) unsafeAtomicAdd(vc + k, v_rgb[k]);
                unsafeAtomicAdd(ga.v_conics + ga.s_conic * g, v_ca);
                unsafeAtomicAdd(ga.v_conics + ga.s_conic * g + 1, v_cb);
                unsafeAtomicAdd(ga.v_conics + ga.s_conic * g + 2, v_cc);
                unsafeAtomicAdd(ga.v_means2d + ga.s_xy * g, v_x);
                unsafeAtomicAdd(ga.v_means2d + ga.s_xy * g + 1, v_y);
                if (ga.v_means2d_abs != nullptr) {
                    unsafeAtomicAdd(ga.v_means2d_abs + ga.s_abs * g, v_ax);
                    unsafeAtomicAdd(ga.v_means2d_abs + ga.s_abs * g + 1, v_ay);
                }
                unsafeAtomicAdd(ga.v_opacities + ga.s_opac * g, v_o);
            }
        }
    }
}

size_t ref_smem_bytes(uint32_t block_size, int cdim, bool bwd) {
    size_t b = 16 * ((block_size * 4 + 15) / 16) + (size_t)block_size * (16 + 8);
    if (bwd) b += (size_t)block_size * cdim * 4;
    return b;
}

} // namespace

template <int CDIM>
static int32_t launch_ref_fwd(const RasterArgs &a, uint32_t cnt, uint32_t ch_off, hipStream_t st) {
    uint32_t block = ((a.tile_size * a.tile_size + 63) / 64) * 64;
    dim3 grid(a.tile_width, a.tile_height, a.C);
    hipLaunchKernelGGL((raster_ref_fwd_kernel<CDIM>), grid, dim3(block), ref_smem_bytes(block, CDIM, false), st, a, cnt, ch_off);
    return 0;
}

template <int CDIM>
static int32_t launch_ref_bwd(const RasterArgs &a, const RasterGradArgs &ga, uint32_t cnt, uint32_t ch_off, int use_va, hipStream_t st) {
    uint32_t block = ((a.tile_size * a.tile_size + 63) / 64) * 64;
    dim3 grid(a.tile_width, a.tile_height, a.C);
    hipLaunchKernelGGL((raster_ref_bwd_kernel<CDIM>), grid, dim3(block), ref_smem_bytes(block, CDIM, true), st, a, ga, cnt, ch_off, use_va);
    return 0;
}

int32_t raster_ref_fwd(const RasterArgs &a, hipStream_t st) {
    if (a.tile_size * a.tile_size > GS_BLOCK) {
        gs_set_error("rasterize (ref impl): tile_size must be <= 16");
        return 1;
    }
    for (uint32_t off = 0; off < a.channels; off += 32) {
        uint32_t cnt = min(32u, a.channels - off);
        if (cnt <= 1) launch_ref_fwd<1>(a, cnt, off, st);
        else if (cnt <= 2) launch_ref_fwd<2>(a, cnt, off, st);
        else if (cnt <= 3) launch_ref_fwd<3>(a, cnt, off, st);
        else if (cnt <= 4) launch_ref_fwd<4>(a, cnt, off, st);
        else if (cnt <= 8) launch_ref_fwd<8>(a, cnt, off, st);
        else if (cnt <= 16) launch_ref_fwd<16>(a, cnt, off, st);
        else launch_ref_fwd<32>(a, cnt, off, st);
    }
    return 0;
}

int32_t raster_ref_bwd(const RasterArgs &a, const RasterGradArgs &ga, hipStream_t st) {
    if (a.tile_size * a.tile_size > GS_BLOCK) {
        gs_set_error("rasterize (ref impl): tile_size must be <= 16");
        return 1;
    }
    for (uint32_t off = 0; off < a.channels; off += 32) {
        uint32_t cnt = min(32u, a.channels - off);
        int use_va = (off == 0) && ga.v_render_alphas != nullptr;
        if (cnt <= 1) launch_ref_bwd<1>(a, ga, cnt, off, use_va, st);
        else if (cnt <= 2) launch_ref_bwd<2>(a, ga, cnt, off, use_va, st);
        else if (cnt <= 3) launch_ref_bwd<3>(a, ga, cnt, off, use_va, st);
        else if (cnt <= 4) launch_ref_bwd<4>(a, ga, cnt, off, use_va, st);
        else if (cnt <= 8) launch_ref_bwd<8>(a, ga, cnt, off, use_va, st);
        else if (cnt <= 16) launch_ref_bwd<16>(a, ga, cnt, off, use_va, st);
        else launch_ref_bwd<32>(a, ga, cnt, off, use_va, st);
    }
    return 0;
}
