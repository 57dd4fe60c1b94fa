// step.hip -- native step driver above the per-operator C ABI (round 4).
//
// One rasterization() forward + backward of the common training case (unpacked batch, quats + scales or covars, shared SH
// coefficients or [N,3] colours, RGB) is ~30 launches through nine operator entry points; driven from Python that is ~0.6 ms
// of host work per step (argument marshalling of seven ctypes calls, two autograd nodes, ~20 tensor allocations) against
// ~0.75 ms of GPU work -- and the multi-GPU modes, which add collectives, are host-bound.  The three functions here issue
// the SAME launches through the SAME entry points from ONE descriptor the caller fills once (`gs_step`: host struct, device
// pointers; every buffer caller-owned, sizes from the gs_*_bytes helpers): the host side of a step becomes three calls.
//   gs_step_fwd_begin   projection into splat rows (+ SH colours) -> splitters -> count + depth keys -> depth pre-sort
//                       [the caller waits for block_sums (pinned), adds them up = n_isects, allocates the phase-2 buffers]
//   gs_step_fwd_finish  emit -> pair sort -> offsets -> compositing forward (+ zero-fill side job)
//   gs_step_bwd         compositing backward -> projection (+ SH) backward
// Nothing is computed here that the operator entry points do not compute: results are identical by construction
// (tests/test_gpu_step.py compares them bit for bit).  The operators stay the drop-in boundary (reference csrc/ext.cpp);
// this is the executor around them, the counterpart of the Python orchestration in gsplat/rendering.py:28-582.
#include "gs_common.h"

#define GS_STEP_TRY(call)            \
    do {                             \
        const int32_t rc_ = (call);  \
        if (rc_ != 0) return rc_;    \
    } while (0)

static uint32_t floor_log2_plus1(uint32_t x) { // floor(log2(x)) + 1, as the reference computes its bit counts (isect_tiles.cu:155-157)
    uint32_t b = 0;
    while (x) {
        ++b;
        x >>= 1;
    }
    return b ? b : 1;
}

#include <cstddef>

extern "C" uint32_t gs_step_layout(uint64_t *out, uint32_t n) {
    const uint64_t v[] = {sizeof(gs_step), offsetof(gs_step, C), offsetof(gs_step, sh_K), offsetof(gs_step, eps2d), offsetof(gs_step, tile_size),
                          offsetof(gs_step, sh_mask_logits), offsetof(gs_step, rows_ready), offsetof(gs_step, backgrounds),
                          offsetof(gs_step, radii), offsetof(gs_step, sort_temp_bytes), offsetof(gs_step, block_sums),
                          offsetof(gs_step, n_isects), offsetof(gs_step, n_kept_host), offsetof(gs_step, work_bytes), offsetof(gs_step, plan), offsetof(gs_step, scratch),
                          offsetof(gs_step, zero_fill_bytes), offsetof(gs_step, v_render_colors), offsetof(gs_step, vrc_pixel_stride),
                          offsetof(gs_step, grad_rows), offsetof(gs_step, v_sh_rest), offsetof(gs_step, absgrad),
                          offsetof(gs_step, finish_phase), offsetof(gs_step, dyn_motion), offsetof(gs_step, dyn_timestamp),
                          offsetof(gs_step, dyn_quant_lo), offsetof(gs_step, v_dyn_motion)};
    const uint32_t m = (uint32_t)(sizeof(v) / sizeof(v[0]));
    for (uint32_t i = 0; out != nullptr && i < n && i < m; ++i) out[i] = v[i];
    return m;
}

extern "C" uint32_t gs_quant_desc_layout(uint64_t *out, uint32_t n) {
    const uint64_t v[] = {sizeof(gs_quant_desc), offsetof(gs_quant_desc, n), offsetof(gs_quant_desc, x), offsetof(gs_quant_desc, out),
                          offsetof(gs_quant_desc, v_out), offsetof(gs_quant_desc, v_x), offsetof(gs_quant_desc, lo),
                          offsetof(gs_quant_desc, q_step), offsetof(gs_quant_desc, activation), offsetof(gs_quant_desc, philox_offset)};
    const uint32_t m = (uint32_t)(sizeof(v) / sizeof(v[0]));
    for (uint32_t i = 0; out != nullptr && i < n && i < m; ++i) out[i] = v[i];
    return m;
}

extern "C" int32_t gs_step_fwd_begin(gs_step *s, gs_stream_t stream) {
    GS_CHECK_ARG(s != nullptr, "null descriptor");
    GS_CHECK_ARG(s->C > 0 && s->N > 0, "C and N must be > 0");
    GS_CHECK_ARG(s->radii && s->depths && s->rows && s->tiles_per_gauss && s->depth_keys && s->sort_temp &&
                     s->perm && s->n_kept && s->group_sums && s->block_sums,
                 "a phase-1 buffer is missing");
    const uint32_t n_elems = s->C * s->N;
    // rows_ready: the splat rows, radii and depths are already there (they arrived through the multi-GPU exchange of the
    // gaussian-sharded mode, C = the local cameras, N = all ranks' splats): binning only, the count kernel counts the tiles
    if (!s->rows_ready && s->dyn_motion != nullptr) {
        GS_CHECK_ARG(s->covars == nullptr && s->sh_coeffs == nullptr, "dynamic splats: quats + scales and [N,3] colours only (no covars, no SH)");
        GS_STEP_TRY(gs_projection_rows_dyn_fwd(s->C, s->N, s->means, const_cast<float *>(s->quats), const_cast<float *>(s->scales), s->dyn_motion,
                                               s->dyn_omega, s->dyn_trbf_center, s->dyn_trbf_scale, s->dyn_timestamp, s->dyn_min_trbf, s->dyn_trbf_alive,
                                               s->dyn_raw_params, s->dyn_quant_mask, s->dyn_quant_lo, s->dyn_quant_hi, s->dyn_quant_range,
                                               s->dyn_quant_step_norm, s->viewmats, s->Ks, s->width, s->height, s->eps2d, s->near_plane,
                                               s->far_plane, s->radius_clip,
                                               s->camera_model, const_cast<float *>(s->opacities), const_cast<float *>(s->colors), s->antialiased,
                                               s->tile_size, s->tile_width, s->tile_height, s->tiles_per_gauss, s->block_sums, s->radii, s->depths,
                                               s->rows, stream));
    } else if (!s->rows_ready)
    GS_STEP_TRY(gs_projection_rows_fwd(s->C, s->N, s->means, s->covars, s->quats, s->scales, s->viewmats, s->Ks, s->width, s->height, s->eps2d,
                                       s->near_plane, s->far_plane, s->radius_clip, s->camera_model, s->opacities, s->colors, s->antialiased,
                                       s->sh_coeffs, s->sh_rest, s->sh_K, s->sh_degree, s->sh_mask_logits, s->sh_mask_temperature, s->sh_mask_binary,
                                       s->tile_size, s->tile_width, s->tile_height, s->tiles_per_gauss, s->block_sums, s->radii, s->depths,
                                       s->rows, stream));
    const int32_t hist_ready = gs_sort_first_hist_applicable(n_elems);
    const bool bucketed = s->bucketed && gs_presort_applicable(n_elems);
    GS_CHECK_ARG(!bucketed || s->splitters != nullptr, "the bucketed pre-sort needs the splitter table");
    GS_CHECK_ARG(bucketed || s->depth_vals != nullptr, "the radix pre-sort needs depth_vals");
    if (bucketed) GS_STEP_TRY(gs_presort_split(n_elems, s->radii, s->depths, s->splitters, stream));
    // (the projection counted the tiles and left the block sums: the count kernel only makes the depth keys and their histogram)
    GS_STEP_TRY(gs_isect_count_keys(n_elems, s->rows_ready ? s->rows + GS_ROW_MEAN2D : nullptr, GS_ROW_FLOATS, s->radii, s->depths, s->tile_size,
                                    s->tile_width, s->tile_height, s->tiles_per_gauss, s->depth_keys, s->depth_vals,
                                    s->rows_ready ? s->block_sums : nullptr, hist_ready ? s->sort_temp : nullptr,
                                    hist_ready ? (size_t)s->sort_temp_bytes : 0, bucketed ? s->splitters : nullptr, stream));
    const uint32_t gshift = gs_isect_emit_group_shift();
    if (bucketed) {
        GS_STEP_TRY(gs_presort_buckets(n_elems, s->depth_keys, s->depth_vals, s->splitters, s->perm, s->n_kept, s->sort_temp,
                                       (size_t)s->sort_temp_bytes, s->tiles_per_gauss, s->group_sums, gshift, s->lds_capacity, stream));
    } else {
        GS_CHECK_ARG(s->sorted_keys != nullptr, "the radix pre-sort needs sorted_keys");
        GS_STEP_TRY(gs_sort_pairs_u64_i32_drop(n_elems, s->depth_keys, s->depth_vals, s->sorted_keys, s->perm, 32, 64, 0x7FFFFFFFu, s->n_kept,
                                               s->sort_temp, (size_t)s->sort_temp_bytes, hist_ready, s->tiles_per_gauss, s->group_sums, gshift,
                                               stream));
    }
    if (s->group_prefix != nullptr) // (many groups: one prefix sum over them instead of a quadratic number of loads in the emission)
        GS_STEP_TRY(gs_cumsum_i32((n_elems + (1u << gshift) - 1) >> gshift, (const int32_t *)s->group_sums, s->group_prefix, s->cumsum_scratch,
                                  (size_t)s->cumsum_scratch_bytes, stream));
    return 0;
}

extern "C" int32_t gs_step_fwd_finish(gs_step *s, gs_stream_t stream) {
    GS_CHECK_ARG(s != nullptr, "null descriptor");
    GS_CHECK_ARG(s->offsets && (s->finish_phase == 1 || (s->render_colors && s->render_alphas && s->last_ids)), "a phase-2 buffer is missing");
    GS_CHECK_ARG(s->n_isects == 0 || (s->isect_ids && s->flatten_ids && s->work), "a phase-2 buffer is missing");
    const uint32_t n_elems = s->C * s->N;
    const uint32_t n_tiles = s->tile_width * s->tile_height;
    // finish_phase 1 / 2: the binning half and the compositing half as separate calls (the caller sizes the compositing
    // scratch from n_isects while the GPU is busy with the binning); 0: both
    if (s->finish_phase != 2)
    GS_STEP_TRY(gs_isect_finish_presorted(n_elems, s->N, s->n_isects, s->perm, s->n_kept, nullptr, s->rows, GS_ROW_FLOATS, s->radii, s->depths,
                                          s->tiles_per_gauss, s->group_sums, s->group_prefix, s->tile_size, s->tile_width, s->tile_height,
                                          floor_log2_plus1(n_tiles), floor_log2_plus1(s->C), s->C, s->isect_ids, s->flatten_ids, s->offsets,
                                          s->work, (size_t)s->work_bytes, s->n_kept_host,
                                          (s->bucketed && gs_presort_applicable(n_elems)) ? s->depth_keys : s->sorted_keys, stream));
    if (s->finish_phase == 1) return 0;
    const uint32_t strides[4] = {GS_ROW_FLOATS, GS_ROW_FLOATS, GS_ROW_FLOATS, GS_ROW_FLOATS};
    GS_STEP_TRY(gs_rasterize_fwd(s->C, n_elems, (uint32_t)s->n_isects, 3, s->rows + GS_ROW_MEAN2D, s->rows + GS_ROW_CONIC, s->rows + GS_ROW_COLOR,
                                 s->rows + GS_ROW_OPACITY, strides, s->backgrounds, nullptr, (uint32_t)s->width, (uint32_t)s->height, s->tile_size,
                                 s->tile_width, s->tile_height, s->offsets, s->flatten_ids, s->render_colors, s->render_alphas, s->last_ids,
                                 s->scratch ? &s->plan : nullptr, s->scratch, s->zero_fill, (size_t)s->zero_fill_bytes, stream));
    return 0;
}

extern "C" int32_t gs_step_bwd(gs_step *s, gs_stream_t stream) {
    GS_CHECK_ARG(s != nullptr, "null descriptor");
    GS_CHECK_ARG(s->grad_rows && s->v_render_colors, "the gradient rows and the image gradient are required");
    const uint32_t n_elems = s->C * s->N;
    const uint32_t strides[4] = {GS_ROW_FLOATS, GS_ROW_FLOATS, GS_ROW_FLOATS, GS_ROW_FLOATS};
    GS_STEP_TRY(gs_rasterize_bwd(s->C, n_elems, (uint32_t)s->n_isects, 3, s->rows + GS_ROW_MEAN2D, s->rows + GS_ROW_CONIC, s->rows + GS_ROW_COLOR,
                                 s->rows + GS_ROW_OPACITY, strides, s->backgrounds, nullptr, (uint32_t)s->width, (uint32_t)s->height, s->tile_size,
                                 s->tile_width, s->tile_height, s->offsets, s->flatten_ids, s->render_colors, s->render_alphas, s->last_ids,
                                 s->v_render_colors, s->v_render_alphas, s->vrc_pixel_stride, s->vrc_channel_stride,
                                 s->absgrad ? s->grad_rows : nullptr, s->grad_rows, nullptr, nullptr, nullptr, 1, nullptr,
                                 s->scratch ? &s->plan : nullptr, s->scratch, stream));
    if (!s->skip_projection_bwd && s->dyn_motion != nullptr)
        GS_STEP_TRY(gs_projection_rows_dyn_bwd(s->C, s->N, s->means, s->quats, s->scales, s->dyn_motion, s->dyn_omega, s->dyn_trbf_center,
                                               s->dyn_trbf_scale, s->dyn_timestamp, s->dyn_raw_params, s->dyn_quant_mask, s->dyn_quant_lo,
                                               s->dyn_quant_hi, s->dyn_quant_range, s->dyn_quant_step_norm, s->viewmats, s->Ks, s->width, s->height,
                                               s->eps2d, s->camera_model, s->radii, s->rows, s->grad_rows, s->v_depths, s->opacities, s->antialiased,
                                               s->v_means, s->v_quats, s->v_scales, s->v_dyn_motion, s->v_dyn_omega, s->v_dyn_trbf_center,
                                               s->v_dyn_trbf_scale, s->v_opacities, s->v_colors, s->outputs_prefilled, stream));
    else if (!s->skip_projection_bwd)
        GS_STEP_TRY(gs_projection_rows_bwd(s->C, s->N, s->means, s->covars, s->quats, s->scales, s->viewmats, s->Ks, s->width, s->height, s->eps2d,
                                           s->camera_model, s->radii, s->rows, s->grad_rows, s->v_depths, s->opacities, s->antialiased, s->v_means,
                                           s->v_covars, s->v_quats, s->v_scales, nullptr, s->v_opacities, s->v_colors, nullptr, s->sh_coeffs,
                                           s->sh_rest, s->sh_K, s->sh_degree, s->v_sh, s->v_sh_rest, s->sh_mask_logits, s->sh_mask_temperature,
                                           s->sh_mask_binary, s->v_sh_mask_logits, s->outputs_prefilled, stream));
    return 0;
}
