/*
 * gsplat_hip.h -- flat C ABI of libgsplat_hip.so (MI355X / gfx950).
 *
 * This is the drop-in boundary for the rasterize + quantize hot path of
 * JasonLSC/GSCodec_Studio (a gsplat 1.4.0 fork).  Every entry point replaces one
 * pybind11 function of the reference's native module (reference file:line cited
 * per function, paths relative to the reference root).  Differences from the
 * reference's native surface, all deliberate:
 *
 *   - no torch types: raw device pointers + explicit sizes + a hipStream_t
 *     (passed as void*), so any host (ctypes, cgo, JNI, a C++ runtime) can bind it;
 *   - the CALLER owns every buffer, including scratch; the library keeps no
 *     global device state and never allocates or frees device memory;
 *   - data-dependent sizes (n_isects, nnz) use a two-call protocol: a *_count
 *     call leaves the total in caller memory, the caller allocates, then calls
 *     the *_emit / *_fill entry point;
 *   - every function returns 0 on success, non-zero on error; the message is
 *     available (thread-local) from gs_last_error().  Nothing throws or aborts
 *     across the ABI;
 *   - all launches are asynchronous on the given stream.
 *
 * All floating-point tensors are fp32, contiguous, row-major.  Integer dtypes
 * match the reference's meta tensors (radii/flatten_ids/offsets/last_ids int32,
 * isect_ids/camera_ids/gaussian_ids int64, masks uint8/bool).
 */
#ifndef GSPLAT_HIP_H
#define GSPLAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Bumped whenever an exported signature or the meaning of an argument changes (1: round 1; 2: the round-2 additions to
 * gs_rasterize_fwd, gs_isect_count_keys, gs_sort_pairs_u64_i32_drop, gs_projection_bwd; 3: round 3 -- the splat-row layout,
 * gs_raster_plan, gs_kmeans_decode's bounds; 4: round 4 -- the shN mask entry points, gs_isect_count_keys' bucket_splitters, the bucketed pre-sort; 5: round 6 -- gs_projection_rows_dyn_*, the dyn_* fields of gs_step, gs_accumulate_*, gs_quantize_round_multi_*, the scratch behind gs_presort_split's table).  A binding must refuse a library whose gs_version() differs from the
 * GS_ABI_VERSION of the header it was generated from, and SHOULD also compare gs_header_hash() (the first 8 bytes of the
 * SHA-256 of the header file the library was compiled against, big-endian) with the hash of its own copy: ctypes / cgo call
 * through shifted argument lists silently otherwise. */
#define GS_ABI_VERSION 5

/* reference: gsplat/cuda/include/bindings.h:34-38 (enum CameraModelType) */
#define GS_CAMERA_PINHOLE 0
#define GS_CAMERA_ORTHO 1
#define GS_CAMERA_FISHEYE 2

/* quantizer modes (reference: gsplat/compression_simulation/ops.py:39-54) */
#define GS_QUANT_NOISE 0
#define GS_QUANT_ROUND 1
/* activation fused behind a quantizer (opt-in; 0 keeps the reference's bit-exact output) */
#define GS_ACT_NONE 0
#define GS_ACT_EXP 1
#define GS_ACT_SIGMOID 2

typedef void *gs_stream_t; /* hipStream_t */

int32_t gs_version(void);
uint64_t gs_header_hash(void);
const char *gs_last_error(void);

/* ------------------------------------------------------------------------
 * Splat rows: ONE 64-byte row of 16 floats per projected splat (camera, gaussian), the layout the compositing kernels
 * fetch with a single L2 request (three 16-byte loads from one line) instead of four gathers from four arrays:
 *   [0] mean2d.x  [1] mean2d.y  [2] conic.a  [3] conic.b  [4] conic.c  [5] opacity (after antialias compensation)
 *   [6] [7] [8] colour            [9] depth   [10] radius (int32 bit pattern)   [11] compensation   [12..15] reserved
 * The reference's tensors (means2d [C,N,2], conics [C,N,3], opacities [C,N], colours [C,N,3]; gsplat/rendering.py:337-349)
 * are COLUMN VIEWS of the row buffer, so `meta` keeps its keys, shapes and dtypes.  gs_projection_rows_fwd fills
 * columns 0-5 and 9-11 (and 6-8 for post-activation colours), gs_sh_view_fwd columns 6-8.  Rows of culled splats
 * (radii == 0) are left untouched, as the reference leaves means2d / conics uninitialised there.  The gradient rows the
 * compositing backward accumulates into (gs_rasterize_bwd, packed16) use the same columns: v_mean2d | v_conic |
 * v_opacity | v_colour (4 channels: column 9 is the fourth) | [10] [11] absgrad.
 * ---------------------------------------------------------------------- */
#define GS_ROW_FLOATS 16
#define GS_ROW_MEAN2D 0
#define GS_ROW_CONIC 2
#define GS_ROW_OPACITY 5
#define GS_ROW_COLOR 6
#define GS_ROW_DEPTH 9
#define GS_ROW_RADIUS 10
#define GS_ROW_COMPENSATION 11


/* ------------------------------------------------------------------------
 * R1  fully fused projection
 * replaces fully_fused_projection_fwd_tensor
 *   (gsplat/cuda/csrc/fully_fused_projection_fwd.cu:198-275, kernel 22-196)
 * Exactly one of {covars} / {quats, scales} is non-NULL.
 * Outputs for entries with radii == 0 are left untouched (the reference leaves
 * them uninitialised); compensations (optional) must be zero-filled by the caller.
 * ---------------------------------------------------------------------- */
int32_t gs_projection_fwd(
    uint32_t C, uint32_t N,
    const float *means,    /* [N,3] */
    const float *covars,   /* [N,6] triu or NULL */
    const float *quats,    /* [N,4] wxyz or NULL */
    const float *scales,   /* [N,3] or NULL */
    const float *viewmats, /* [C,4,4] world->camera */
    const float *Ks,       /* [C,3,3] */
    int32_t image_width, int32_t image_height,
    float eps2d, float near_plane, float far_plane, float radius_clip,
    int32_t camera_model,
    int32_t *radii,       /* [C,N] */
    float *means2d,       /* [C,N,2] */
    float *depths,        /* [C,N] */
    float *conics,        /* [C,N,3] */
    float *compensations, /* [C,N] or NULL */
    gs_stream_t stream);

/* replaces fully_fused_projection_bwd_tensor
 *   (gsplat/cuda/csrc/fully_fused_projection_bwd.cu:265-372, kernel 24-263)
 * v_means / v_covars / v_quats / v_scales are fully OVERWRITTEN (one thread owns one
 * gaussian and sums over cameras in registers: no atomics, deterministic).
 * v_viewmats (optional) is ACCUMULATED with atomics: caller zero-fills it. */
int32_t gs_projection_bwd(
    uint32_t C, uint32_t N,
    const float *means, const float *covars, const float *quats, const float *scales,
    const float *viewmats, const float *Ks,
    int32_t image_width, int32_t image_height, float eps2d, int32_t camera_model,
    const int32_t *radii,        /* [C,N] */
    const float *conics,         /* [C,N,3] */
    const float *compensations,  /* [C,N] or NULL */
    const float *v_means2d,      /* [C,N,2] */
    const float *v_depths,       /* [C,N] or NULL (= zero) */
    const float *v_conics,       /* [C,N,3] */
    const float *v_compensations,/* [C,N] or NULL */
    float *v_means,   /* [N,3] or NULL */
    float *v_covars,  /* [N,6] or NULL */
    float *v_quats,   /* [N,4] or NULL */
    float *v_scales,  /* [N,3] or NULL */
    float *v_viewmats,/* [C,4,4] or NULL */
    uint32_t v_means2d_stride, /* row stride of v_means2d in floats: 2, or 16 for the packed compositing rows */
    uint32_t v_conics_stride,  /* row stride of v_conics in floats: 3, or 16 */
    const float *v_means_add,  /* [N,3] or NULL: added to v_means (the d/d means that arrives through the SH view
                                  directions, rendering.py:381-391 in the reference; saves autograd's elementwise sum) */
    gs_stream_t stream);

/* Row form of the fused projection (what rasterization() uses for unpacked batches): same arithmetic and culling as
 * gs_projection_fwd, but every visible (camera, gaussian) pair gets ONE splat row (see "Splat rows" above) instead of
 * entries in three arrays.  Folded in, because the row wants them and the pass runs anyway:
 *   opacities [N] (or NULL): column 5 = opacities[n]              -- the `opacities.repeat(C, 1)` of rendering.py:331,
 *                            x compensation when antialiased != 0  -- and the multiply of rendering.py:334-335;
 *   colors [N,3] (or NULL):  columns 6-8 = colors[n]               -- the `colors.expand(C, -1, -1)` of rendering.py:386;
 *   sh_coeffs [N,sh_K,3] (or NULL; excludes colors): columns 6-8 = max(SH(means[n] - camera centre) + 0.5, 0) for the
 *                            first (sh_degree + 1)^2 bands -- exactly what gs_sh_view_fwd writes (same arithmetic, bit for
 *                            bit), without its launch and its second pass over means / radii; its gradient stays with
 *                            gs_sh_view_bwd.
 * radii [C,N] and depths [C,N] are ALSO written densely (the binning kernels stream through them); rows must be 64-byte
 * aligned; rows / depths of culled pairs are left untouched. */
int32_t gs_projection_rows_fwd(
    uint32_t C, uint32_t N,
    const float *means, const float *covars, const float *quats, const float *scales,
    const float *viewmats, const float *Ks,
    int32_t image_width, int32_t image_height,
    float eps2d, float near_plane, float far_plane, float radius_clip,
    int32_t camera_model,
    const float *opacities, const float *colors, int32_t antialiased,
    const float *sh_coeffs, const float *sh_coeffs_rest /* NULL or split rows, as in gs_sh_view_fwd */, uint32_t sh_K,
    uint32_t sh_degree,
    const float *sh_mask_logits /* NULL, or [N] (split rows only): the shN mask of the compression-simulation hooks
                                   (gs_shn_mask_fwd) applied on the fly -- the coefficients of the bands >= 1 are
                                   multiplied by sigmoid(logit / temperature) (binary: sigmoid(logit) >= 0.5) as they are
                                   loaded, the masked coefficients are never materialised; bit-identical to masking first */,
    float sh_mask_temperature, int32_t sh_mask_binary,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    int32_t *tiles_per_gauss /* NULL, or [C,N]: the binning's tile count of every pair (gs_isect_count), in the same pass */,
    int32_t *block_sums /* NULL, or [C * gs_projection_rows_blocks(N)][2] (8-byte aligned; may be pinned host memory): per
                           workgroup (sum of the counts, number of visible pairs), written as ONE 8-byte store -- the totals
                           are n_isects and the number of elements the depth pre-sort keeps, known to the host one kernel into
                           the step (the read-back of isect_tiles.cu:200) while the whole depth pre-sort is still queued;
                           gs_isect_count_keys then takes means2d = NULL, block_sums = NULL (tiles_per_gauss is an input there) */,
    int32_t *radii, /* [C,N] */
    float *depths,  /* [C,N] */
    float *rows,    /* [C,N,16] */
    gs_stream_t stream);
uint32_t gs_projection_rows_blocks(uint32_t N);
/* Its backward: grad_rows [C,N,16] holds d/d(mean2d, conic, opacity, colour) in the splat-row columns (what gs_rasterize_bwd
 * accumulates with packed16), v_depths [C,N] or NULL.  Besides the outputs of gs_projection_bwd (all OVERWRITTEN),
 * v_opacities [N] = sum over cameras of column 5 (x compensation when antialiased, whose own gradient then enters the
 * projection chain) and v_colors [N,3] = sum over cameras of columns 6-8; either may be NULL.
 * outputs_prefilled != 0: the caller guarantees that every per-gaussian output already holds zeros (e.g. through the
 * zero_fill of gs_rasterize_fwd); the rows of gaussians that no camera sees (71 % at BASELINE config 2) are then not
 * written at all, and v_means_add is only read where some camera sees the gaussian.
 * sh_coeffs != NULL (the forward evaluated the colours from them): the SH backward of gs_sh_view_bwd runs IN THE SAME PASS --
 * v_sh_coeffs [N,K,3] (or the pair v_sh_coeffs [N,1,3] / v_sh_coeffs_rest [N,K-1,3] for split coefficients) is written, and
 * the gradient that reaches the means through the view directions is added to v_means without leaving the lane.  Needs
 * 3 K % 4 == 0, 16-byte aligned coefficient / gradient rows, v_viewmats == NULL and v_colors == NULL; otherwise call
 * gs_sh_view_bwd first and hand its v_means over as v_means_add. */
int32_t gs_projection_rows_bwd(
    uint32_t C, uint32_t N,
    const float *means, const float *covars, const float *quats, const float *scales,
    const float *viewmats, const float *Ks,
    int32_t image_width, int32_t image_height, float eps2d, int32_t camera_model,
    const int32_t *radii, const float *rows, const float *grad_rows, const float *v_depths,
    const float *opacities /* [N]; needed when antialiased */, int32_t antialiased,
    float *v_means, float *v_covars, float *v_quats, float *v_scales, float *v_viewmats,
    float *v_opacities, float *v_colors,
    const float *v_means_add, /* [N,3] or NULL, as in gs_projection_bwd */
    const float *sh_coeffs, const float *sh_coeffs_rest, uint32_t sh_K, uint32_t sh_degree,
    float *v_sh_coeffs, float *v_sh_coeffs_rest,
    const float *sh_mask_logits, float sh_mask_temperature, int32_t sh_mask_binary, /* the forward's mask (or NULL, 0, 0) */
    float *v_sh_mask_logits /* [N] or NULL: gradient of the logits, every entry written (training-mode mask only) */,
    int32_t outputs_prefilled,
    gs_stream_t stream);

/* packed (COO) projection, replaces fully_fused_projection_packed_fwd_tensor
 *   (gsplat/cuda/csrc/fully_fused_projection_packed_fwd.cu:251-399).
 * Pass 1 (count): block_cnts[C * nblocks] <- #visible per (camera, 256-gaussian block),
 * nblocks = ceil(N / 256).  The caller turns it into an inclusive prefix sum
 * (gs_cumsum_i32) and reads nnz = last element.  Pass 2 (fill) writes rows in
 * (camera, gaussian) order.  NOTE the packed radius formula differs from the unpacked
 * one exactly as in the reference (packed_fwd.cu:183-186 vs fwd.cu:167-169). */
int32_t gs_projection_packed_count(
    uint32_t C, uint32_t N,
    const float *means, const float *covars, const float *quats, const float *scales,
    const float *viewmats, const float *Ks,
    int32_t image_width, int32_t image_height,
    float eps2d, float near_plane, float far_plane, float radius_clip,
    int32_t camera_model,
    int32_t *block_cnts, /* [C * ceil(N/256)] */
    gs_stream_t stream);

int32_t gs_projection_packed_fill(
    uint32_t C, uint32_t N,
    const float *means, const float *covars, const float *quats, const float *scales,
    const float *viewmats, const float *Ks,
    int32_t image_width, int32_t image_height,
    float eps2d, float near_plane, float far_plane, float radius_clip,
    int32_t camera_model,
    const int32_t *block_accum, /* inclusive prefix sum of block_cnts */
    int32_t *indptr,            /* [C+1] */
    int64_t *camera_ids,        /* [nnz] */
    int64_t *gaussian_ids,      /* [nnz] */
    int32_t *radii, float *means2d, float *depths, float *conics,
    float *compensations,       /* [nnz] or NULL */
    gs_stream_t stream);

/* replaces fully_fused_projection_packed_bwd_tensor
 *   (gsplat/cuda/csrc/fully_fused_projection_packed_bwd.cu:318-430).
 * sparse_grad == 0: dense [N,*] outputs, ACCUMULATED with atomics (caller zero-fills);
 * sparse_grad != 0: [nnz,*] outputs written directly. */
int32_t gs_projection_packed_bwd(
    uint32_t C, uint32_t N, uint32_t nnz,
    const float *means, const float *covars, const float *quats, const float *scales,
    const float *viewmats, const float *Ks,
    int32_t image_width, int32_t image_height, float eps2d, int32_t camera_model,
    const int64_t *camera_ids, const int64_t *gaussian_ids,
    const float *conics, const float *compensations,
    const float *v_means2d, const float *v_depths, const float *v_conics,
    const float *v_compensations,
    int32_t sparse_grad,
    float *v_means, float *v_covars, float *v_quats, float *v_scales, float *v_viewmats,
    gs_stream_t stream);

/* ------------------------------------------------------------------------
 * R2  spherical harmonics
 * replaces compute_sh_fwd_tensor / compute_sh_bwd_tensor
 *   (gsplat/cuda/csrc/compute_sh_fwd.cu:40-72, compute_sh_bwd.cu:53-95,
 *    gsplat/cuda/include/spherical_harmonics.cuh:13-362)
 * Elements are laid out [C, N]; coeffs is [C,N,K,3] (coeffs_shared == 0) or
 * [N,K,3] broadcast over cameras (coeffs_shared != 0) -- the broadcast form avoids
 * the reference's [C,N,K,3] materialisation (gsplat/cuda/_wrapper.py:72).
 * masks: optional uint8 [C,N]; colours of masked-out elements are left untouched.
 * bwd: v_coeffs is fully OVERWRITTEN (zeros for masked / inactive bases),
 * shape [N,K,3] when coeffs_shared (sum over cameras done in-kernel) else [C,N,K,3].
 * v_dirs (optional, [C,N,3]) is fully overwritten.
 * ---------------------------------------------------------------------- */
int32_t gs_sh_fwd(
    uint32_t C, uint32_t N, uint32_t K, uint32_t degree,
    const float *dirs, const float *coeffs, int32_t coeffs_shared,
    const uint8_t *masks, float *colors, gs_stream_t stream);

int32_t gs_sh_bwd(
    uint32_t C, uint32_t N, uint32_t K, uint32_t degree,
    const float *dirs, const float *coeffs, int32_t coeffs_shared,
    const uint8_t *masks, const float *v_colors,
    float *v_coeffs, float *v_dirs, gs_stream_t stream);

/* Fused "view" form used by rasterization() for coefficients shared by all cameras: the torch
 * ops around the reference's spherical_harmonics call (gsplat/rendering.py:372-392) are folded in:
 *   dirs = means[n] - campos[c]   (campos = inverse(viewmats)[:, :3, 3], [C,3]; with campos_from_viewmats != 0 the
 *                                  pointer holds the [C,4,4] world->camera matrices and the centre is derived in-kernel)
 *   mask = radii[c,n] > 0         (radii may be NULL: no mask)
 *   colors = max(SH + 0.5, 0)
 * bwd: colors_out is the forward output (gradient of the clamp); v_colors may be a strided view
 * (row stride v_colors_stride floats, e.g. 16 for the packed compositing gradient rows);
 * v_coeffs [N,K,3] and v_means [N,3] (= sum over cameras of d/d dirs; may be NULL) are OVERWRITTEN.
 * Optional riders (both NULL to disable): fwd writes opacities_cn[c,n] = opacities[n] for EVERY element (the
 * `opacities.repeat(C, 1)` of rendering.py:331); bwd writes v_opacities[n] = sum_c v_opacities_cn[c,n] (row stride
 * v_opacities_stride floats) -- the two extra torch kernels of the pipeline disappear into passes that run anyway.
 * bwd, outputs_prefilled != 0: the caller guarantees v_coeffs (and v_coeffs_rest) already hold zeros; the rows of
 * gaussians no camera sees are then not written (137 of the 193 MB at BASELINE config 2), and v_means is only DEFINED
 * for gaussians some camera sees. */
/* campos[c] = inverse(viewmats[c])[:3, 3] for affine world->camera matrices, closed form
 * (replaces torch.inverse(viewmats) of gsplat/rendering.py:370, which host-synchronises on ROCm). */
int32_t gs_camera_centers(uint32_t C, const float *viewmats, float *campos, gs_stream_t stream);
int32_t gs_sh_view_fwd(
    uint32_t C, uint32_t N, uint32_t K, uint32_t degree,
    const float *means, const float *campos, int32_t campos_from_viewmats, const float *coeffs,
    const float *coeffs_rest /* NULL, or SPLIT rows: coeffs is the DC band [N,1,3] and this the higher bands [N,K-1,3] -- the
                                trainer's sh0 / shN parameters taken as they are, without the torch.cat of
                                examples/simple_trainer.py:779-786 (193 MB each way at 1 M splats) */,
    const int32_t *radii,
    float *colors, uint32_t colors_stride /* row stride in floats: 3, or GS_ROW_FLOATS when `colors` is column GS_ROW_COLOR of the splat rows */,
    const float *opacities /* [N] or NULL */, float *opacities_cn /* [C,N] or NULL */, gs_stream_t stream);
int32_t gs_sh_view_bwd(
    uint32_t C, uint32_t N, uint32_t K, uint32_t degree,
    const float *means, const float *campos, int32_t campos_from_viewmats, const float *coeffs, const float *coeffs_rest,
    const int32_t *radii,
    const float *colors_out, uint32_t colors_out_stride, const float *v_colors, uint32_t v_colors_stride,
    float *v_coeffs, float *v_coeffs_rest /* [N,K-1,3] with coeffs_rest, else NULL */, float *v_means,
    const float *v_opacities_cn /* or NULL */, uint32_t v_opacities_stride, float *v_opacities /* [N] or NULL */,
    int32_t outputs_prefilled,
    gs_stream_t stream);

/* ------------------------------------------------------------------------
 * R3 / R4  tile intersection, 64-bit radix sort, offset encode
 * replaces isect_tiles_tensor / isect_offset_encode_tensor
 *   (gsplat/cuda/csrc/isect_tiles.cu:106-306, 356-389)
 * Protocol: gs_isect_count -> gs_cumsum_i32 (inclusive, int64 out; last element
 * = n_isects) -> caller allocates -> gs_isect_emit -> gs_sort_pairs_u64_i32 ->
 * gs_isect_offset_encode.
 * ---------------------------------------------------------------------- */
int32_t gs_isect_count(
    uint32_t n_elems,           /* C*N or nnz */
    const float *means2d, uint32_t means2d_stride /* row stride in floats: 2, or GS_ROW_FLOATS for the splat rows */,
    const int32_t *radii,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    int32_t *tiles_per_gauss,   /* [n_elems] */
    gs_stream_t stream);

/* inclusive prefix sum int32 -> int64.  scratch: gs_cumsum_scratch_bytes(n). */
size_t gs_cumsum_scratch_bytes(uint64_t n);
int32_t gs_cumsum_i32(
    uint64_t n, const int32_t *in, int64_t *out,
    void *scratch, size_t scratch_bytes, gs_stream_t stream);
/* int32 -> int32 variant (packed projection block counts). */
int32_t gs_cumsum_i32_i32(
    uint64_t n, const int32_t *in, int32_t *out,
    void *scratch, size_t scratch_bytes, gs_stream_t stream);

/* Splat-level depth pre-sort (an optimisation of the sorted path, results identical):
 * keys[i] = float_bits(depth_i) << 32 | i for visible elements, a maximal key for culled ones.
 * Sorting them (gs_sort_pairs_u64_i32, bits [32,64)) gives a permutation `perm` in (depth, index)
 * order; emitting the intersections in that order (gs_isect_emit with perm, cum_tiles being the
 * prefix sum of gs_gather_i32(tiles_per_gauss, perm)) leaves only the (camera, tile) bits
 * [32, 32+tile_bits+cam_bits) to be sorted afterwards -- 2 radix passes over the n_isects pairs
 * instead of 6 -- and, the radix sort being stable, yields exactly the order of a full-key sort
 * of the reference's emission order (ties: ascending flatten id). */
int32_t gs_isect_depth_keys(
    uint32_t n_elems, const int32_t *radii, const float *depths,
    int64_t *keys, int32_t *vals, gs_stream_t stream);
int32_t gs_gather_i32(uint32_t n, const int32_t *src, const int32_t *idx, int32_t *out, gs_stream_t stream);
/* the same in fewer launches: gs_isect_count + gs_isect_depth_keys in one kernel, and the inclusive prefix sum
 * of in[idx[i]] (= gs_gather_i32 followed by gs_cumsum_i32) without the intermediate array */
int32_t gs_isect_count_keys(
    uint32_t n_elems, const float *means2d /* NULL: tiles_per_gauss is an INPUT (gs_projection_rows_fwd counted) */, uint32_t means2d_stride, const int32_t *radii, const float *depths,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    int32_t *tiles_per_gauss, int64_t *keys, int32_t *vals /* element indices; may be NULL with bucket_splitters (the bucketed
                           pre-sort takes the element from the key's low half) */,
    int32_t *block_sums /* [gs_isect_count_blocks(n_elems)][2] (8-byte aligned) or NULL: per block (intersections, visible
                           elements), ONE 8-byte store; the totals are n_isects -- known here, before the depth pre-sort and the
                           prefix sum, so the host read-back of isect_tiles.cu:200 can overlap them -- and the number of
                           elements the pre-sort keeps (gs_isect_finish_presorted's n_kept_host) */,
    void *sort_temp, size_t sort_temp_bytes /* NULL, 0 -- or the temp buffer the keys will be sorted with
                           (gs_sort_pairs_u64_i32_drop over bits [32, 64)), when gs_sort_first_hist_applicable(n_elems): this
                           kernel then also counts the digits of that sort's first pass (pass first_hist_ready = 1 there) */,
    const int64_t *bucket_splitters /* NULL, or the table of gs_presort_split (with sort_temp = the temp buffer of
                           gs_presort_buckets): the histogram is then counted per BUCKET, for the bucketed pre-sort below */,
    gs_stream_t stream);
uint32_t gs_isect_count_blocks(uint32_t n_elems);
int32_t gs_cumsum_gather_i32(
    uint64_t n, const int32_t *in, const int32_t *idx, const uint32_t *n_valid /* device scalar or NULL: only out[0, *n_valid) is defined (hand the same n_valid to gs_isect_emit*) */,
    int64_t *out, void *scratch, size_t scratch_bytes, gs_stream_t stream);

int32_t gs_isect_emit(
    uint32_t n_elems, uint32_t N,   /* camera of element i = i / N when camera_ids NULL */
    const int32_t *perm,            /* [n_elems] emission order or NULL (identity) */
    const uint32_t *n_valid,        /* device scalar or NULL: only perm[0 .. *n_valid) is defined (gs_sort_pairs_u64_i32_drop) */
    const int64_t *camera_ids,      /* [nnz] or NULL */
    const float *means2d, uint32_t means2d_stride, const int32_t *radii, const float *depths,
    const int64_t *cum_tiles_per_gauss, /* inclusive, indexed by emission position */
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    uint32_t tile_n_bits,
    int64_t *isect_ids,   /* [n_isects] */
    int32_t *flatten_ids, /* [n_isects] */
    gs_stream_t stream);

/* The sorted path in compact form (what rasterization() uses): gs_isect_emit_compact writes every (tile, splat) pair as a
 * 32-bit key  camera << tile_n_bits | tile  plus the flatten id (8 bytes instead of 12), in the depth order of `perm`;
 * gs_sort_isect_pairs sorts the pairs stably by the low `key_bits` (= tile_n_bits + cam_n_bits) bits of the key and its
 * LAST pass writes the reference's outputs: isect_ids = key << 32 | float_bits(depths[flatten id]) and flatten_ids --
 * bit-identical to gs_isect_emit + gs_sort_pairs_u64_i32 (isect_tiles.cu:89-103, 245-299).  keys32 / vals are scratch:
 * destroyed.  temp: gs_sort_isect_temp_bytes(n). */
int32_t gs_isect_emit_compact(
    uint32_t n_elems, uint32_t N, const int32_t *perm, const uint32_t *n_valid, const int64_t *camera_ids,
    const float *means2d, uint32_t means2d_stride, const int32_t *radii, const float *depths, const int64_t *cum_tiles_per_gauss,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, uint32_t tile_n_bits,
    uint32_t *keys32,     /* [n_isects] */
    int32_t *flatten_ids, /* [n_isects] */
    gs_stream_t stream);
/* The same emission with the prefix scan folded in, for the depth-pre-sorted path (perm from gs_sort_pairs_u64_i32_drop):
 * instead of cum_tiles_per_gauss it takes tiles_per_gauss (indexed by element) and group_sums[g] = the tiles of the emission
 * positions [g << s, (g + 1) << s), s = gs_isect_emit_group_shift() -- what that sort's last pass leaves behind in its
 * side_sums (side_vals = tiles_per_gauss, side_shift = s).  compact != 0: keys32 as gs_isect_emit_compact; else isect_ids. */
uint32_t gs_isect_emit_group_shift(void);
uint32_t gs_isect_emit_prefix_from_groups(void); /* more groups than this: hand group_prefix (gs_cumsum_i32 of group_sums) */
int32_t gs_isect_emit_presorted(
    uint32_t n_elems, uint32_t N, const int32_t *perm, const uint32_t *n_valid, const int64_t *camera_ids,
    const float *means2d, uint32_t means2d_stride, const int32_t *radii, const float *depths,
    const int32_t *tiles_per_gauss, const uint32_t *group_sums,
    const int64_t *group_prefix /* NULL, or the inclusive prefix sum of group_sums (gs_cumsum_i32): every workgroup adds up
                                   its predecessors' sums itself otherwise -- fine up to a few thousand groups */,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, uint32_t tile_n_bits, int32_t compact,
    int64_t *isect_ids, uint32_t *keys32, int32_t *flatten_ids, gs_stream_t stream);
/* The whole second half of the sorted binning in ONE call (what rasterization() issues right after its host read-back of
 * n_isects, while the GPU still works off the depth pre-sort: three entry points' worth of host work in one):
 * gs_isect_emit_presorted(compact) -> gs_sort_isect_pairs -> gs_isect_offset_encode.  work: caller-allocated,
 * gs_isect_finish_work_bytes(n_isects) bytes, 16-byte aligned (the compact pairs and the sort's temporaries). */
size_t gs_isect_finish_work_bytes(uint64_t n_isects);
int32_t gs_isect_finish_presorted(
    uint32_t n_elems, uint32_t N, uint64_t n_isects, const int32_t *perm, const uint32_t *n_valid, const int64_t *camera_ids,
    const float *means2d, uint32_t means2d_stride, const int32_t *radii, const float *depths,
    const int32_t *tiles_per_gauss, const uint32_t *group_sums, const int64_t *group_prefix,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, uint32_t tile_n_bits, uint32_t cam_n_bits, uint32_t C,
    int64_t *isect_ids /* [n_isects] */, int32_t *flatten_ids /* [n_isects] */, int32_t *offsets /* [C, tile_height, tile_width] */,
    void *work, size_t work_bytes,
    uint32_t n_kept_host /* 0, or the number of elements in perm (*n_valid) when the host knows it (the block sums of
                            gs_projection_rows_fwd / gs_isect_count_keys): when (camera, tile) key and emission position fit 32
                            bits together the pairs then travel PACKED -- gs_isect_emit_packed + gs_sort_isect_packed, 4 bytes
                            per pair instead of 8 through emission and sort; same outputs */,
    const int64_t *sorted_keys /* NULL, or the pre-sort's sorted keys (depth bits << 32 | element, in perm's order: keys_out of
                            gs_sort_pairs_u64_i32_drop, keys_in of gs_presort_buckets after the call): the packed route's last pass
                            then fetches flatten id and depth bits with one gather */,
    gs_stream_t stream);
/* Packed pairs.  A pair is ONE 32-bit word  (camera << tile_n_bits | tile) << pos_bits | emission position, the position being
 * the splat's index in perm = its depth rank: the words of one tile are in depth order when the array is sorted stably on the
 * key bits, so no value array rides along, and the last pass writes the reference's outputs through perm:
 * flatten_ids = perm[position], isect_ids = key << 32 | float_bits(depths[flatten id]).  Needs key_bits (the bits a key can
 * have set, <= 31) + pos_bits (2^pos_bits >= number of elements in perm) <= 32.  words: scratch, destroyed.
 * temp: gs_sort_isect_temp_bytes(n). */
int32_t gs_isect_emit_packed(
    uint32_t n_elems, uint32_t N, const int32_t *perm, const uint32_t *n_valid, const int64_t *camera_ids,
    const float *means2d, uint32_t means2d_stride, const int32_t *radii, const float *depths,
    const int32_t *tiles_per_gauss, const uint32_t *group_sums, const int64_t *group_prefix,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, uint32_t tile_n_bits, uint32_t pos_bits,
    uint32_t *words /* [n_isects] */, gs_stream_t stream);
int32_t gs_sort_isect_packed(
    uint64_t n, uint32_t *words, const int32_t *perm, const int64_t *sorted_keys /* NULL, or (depth bits << 32 | element) by
    position: replaces perm + depths */, const float *depths /* indexed by the flatten id */, int32_t key_bits,
    uint32_t pos_bits, int64_t *isect_ids, int32_t *flatten_ids, void *temp, size_t temp_bytes, gs_stream_t stream);
size_t gs_sort_isect_temp_bytes(uint64_t n);
int32_t gs_sort_isect_pairs(
    uint64_t n, uint32_t *keys32, int32_t *vals, const float *depths /* indexed by the flatten id */, int32_t key_bits,
    int64_t *isect_ids, int32_t *flatten_ids, void *temp, size_t temp_bytes, gs_stream_t stream);

/* Stable LSD radix sort of (uint64 key, int32 value) pairs on key bits
 * [begin_bit, end_bit) -- the replacement for cub::DeviceRadixSort::SortPairs
 * (gsplat/cuda/csrc/isect_tiles.cu:245-299).  Inputs are not modified; the sorted
 * result is in keys_out / vals_out. */
size_t gs_sort_temp_bytes(uint64_t n);
int32_t gs_sort_pairs_u64_i32(
    uint64_t n,
    const int64_t *keys_in, const int32_t *vals_in,
    int64_t *keys_out, int32_t *vals_out,
    int32_t begin_bit, int32_t end_bit,
    void *temp, size_t temp_bytes, gs_stream_t stream);
/* Same sort, but keys whose upper 32 bits equal drop_hi32 are DROPPED by the first pass (not counted, not written):
 * the outputs hold the *n_kept surviving pairs in sorted order (n_kept: device scalar, written by the call), the rest
 * of the output arrays is untouched.  Used for the splat-level depth pre-sort, where culled splats carry the maximal
 * key and would otherwise be moved through every pass (71% of the elements at BASELINE config 2). */
int32_t gs_sort_pairs_u64_i32_drop(
    uint64_t n, const int64_t *keys_in, const int32_t *vals_in, int64_t *keys_out, int32_t *vals_out,
    int32_t begin_bit, int32_t end_bit, uint32_t drop_hi32, uint32_t *n_kept,
    void *temp, size_t temp_bytes,
    int32_t first_hist_ready /* != 0: the first pass's block histogram is already in temp (gs_isect_count_keys) */,
    const int32_t *side_vals, uint32_t *side_sums, uint32_t side_shift /* both NULL, or: the LAST pass also leaves
                             side_sums[g] = sum of side_vals[vals_out[p]] over the output positions p in [g << side_shift,
                             (g + 1) << side_shift), for all ceil(n / 2^side_shift) groups -- the block sums of the prefix scan
                             gs_isect_emit_presorted needs (side_vals = tiles_per_gauss), with no launch of their own */,
    gs_stream_t stream);
int32_t gs_sort_first_hist_applicable(uint64_t n);

/* Bucketed depth pre-sort (round 4): the same result as gs_sort_pairs_u64_i32_drop over bits [32, 64) with drop_hi32 =
 * 0x7fffffff on the keys of gs_isect_count_keys -- perm = the elements in (depth bits, element) order, culled ones dropped --
 * in 4 launches instead of 11 for the sizes where every radix launch sits at its latency floor (gs_presort_applicable(n):
 * n <= 2 M elements):
 *   gs_presort_split     255 splitters at equal ranks among the visible ones of 8192 elements sampled at a regular stride
 *                        (round 6: individual elements fetched by 8 workgroups instead of 512 runs of 16 by one -- a spatially
 *                        sorted splat array made the runs' depths nearly equal and the buckets overflow).  splitters: int64
 *                        [gs_presort_split_elems()]: the 256-entry table in front, the call's candidate slots behind it (no
 *                        initialisation needed; two launches: 32 sampling workgroups, then the splitter workgroup)
 *   gs_isect_count_keys  (bucket_splitters = the table, sort_temp = temp) counts every 1024-element block's keys per bucket
 *   gs_presort_buckets   scan + ONE stable partition pass by bucket + local sorts: workgroup w finishes the buckets starting in
 *                        positions [1024 w, 1024 (w + 1)) with an LSD sort in LDS on the depth bits that differ inside its range.
 * A range above lds_capacity keys (0 = gs_presort_capacity() = 4096; smaller values are for tests) is sorted by its workgroup
 * through global memory: slower, same result -- with sampled splitters that takes adversarial input.
 * temp: gs_presort_temp_bytes(n).  side_vals / side_sums / side_shift: as in gs_sort_pairs_u64_i32_drop (side_sums is zeroed
 * by the call). */
int32_t gs_presort_applicable(uint64_t n);
uint32_t gs_presort_capacity(void);
size_t gs_presort_temp_bytes(uint64_t n);
uint32_t gs_presort_split_elems(void);
int32_t gs_presort_split(
    uint32_t n_elems, const int32_t *radii, const float *depths, int64_t *splitters /* [gs_presort_split_elems()], see above */,
    gs_stream_t stream);
int32_t gs_presort_buckets(
    uint64_t n, int64_t *keys_in /* DESTROYED: on return [0, *n_kept) holds the sorted keys (depth bits << 32 | element), what
    gs_isect_finish_presorted takes as sorted_keys */, const int32_t *vals_in /* unused (may be NULL) */, const int64_t *splitters,
    int32_t *perm /* [n]; [0, *n_kept) written */, uint32_t *n_kept /* device scalar, written */,
    void *temp, size_t temp_bytes, const int32_t *side_vals, uint32_t *side_sums, uint32_t side_shift,
    uint32_t lds_capacity, gs_stream_t stream);

int32_t gs_isect_offset_encode(
    uint32_t n_isects, const int64_t *isect_ids_sorted,
    uint32_t C, uint32_t n_tiles, uint32_t tile_n_bits,
    int32_t *offsets, /* [C, n_tiles]; fully written (zeros when n_isects == 0) */
    gs_stream_t stream);

/* ------------------------------------------------------------------------
 * R5  per-tile alpha compositing
 * replaces rasterize_to_pixels_fwd_tensor / rasterize_to_pixels_bwd_tensor
 *   (gsplat/cuda/csrc/rasterize_to_pixels_fwd.cu:187-352, kernel 16-185;
 *    gsplat/cuda/csrc/rasterize_to_pixels_bwd.cu:279-489, kernel 17-277)
 * channels is a runtime value (1..513); no padding is required from the caller.
 * n_elems = C*N (unpacked) or nnz (packed): size of the per-splat arrays.
 * bwd outputs are ACCUMULATED with atomics: caller zero-fills them.
 * Channel counts: 1..4 and 5..16 run as one launch each way (tile forward + depth-segmented backward, colours in the LDS
 * records), 17..32 as two launches over halves of the channels (the gradients are linear in v_render_colors; with absgrad,
 * which is not, the backward of more than 16 channels takes the generic one-pass kernel), more than 32 the generic kernels.
 * scratch (optional, gs_raster_plan.scratch_bytes bytes): the forward stores per-pixel checkpoints (transmittance, accumulated colour) at fixed
 * list-index boundaries in it; when the SAME buffer (contents preserved) and the forward's
 * render_colors are handed to gs_rasterize_bwd, the backward runs depth-segmented (one wave per
 * (tile, segment), no serial walk of long lists).  Without them it falls back to one wave per
 * quadrant.  Results are the same up to fp32 rounding.
 * ---------------------------------------------------------------------- */
/* The launch plan of one gs_rasterize_fwd / gs_rasterize_bwd pair: a HOST struct the caller owns, filled by
 * gs_rasterize_plan and handed to both calls, so the two cannot disagree about the scratch layout and the library
 * keeps no state of its own (no process-global tuning).  tuning: NULL (the measured MI355X optima) or 4 ints
 *   [0] segment length of the depth-segmented backward in list entries (multiple of 64; 0 = unsegmented; default 256)
 *   [1] list length from which a tile's four forward waves stop cooperating (0 = never; default 2048)
 *   [2] / [3] work items per XCD group of the forward / backward (0 = identity mapping; default 16)
 * a negative entry keeps the default.  Results never depend on the plan beyond floating-point association of the
 * gradient atomics.  scratch_bytes: size of the scratch buffer both calls take (forward checkpoints, the backward's
 * work list). */
typedef struct gs_raster_plan {
    uint32_t magic, n_tiles_all, n_isects, channels;
    int32_t seg, solo_min;
    uint32_t xcd_fwd, xcd_bwd;
    uint64_t scratch_bytes;
    uint32_t reserved[6];
} gs_raster_plan;
int32_t gs_rasterize_plan(uint32_t n_tiles_all /* C * tile_width * tile_height */, uint32_t n_isects, uint32_t channels,
                          const int32_t *tuning /* [5] HOST ints or NULL */, gs_raster_plan *plan);

int32_t gs_rasterize_fwd(
    uint32_t C, uint32_t n_elems, uint32_t n_isects, uint32_t channels,
    const float *means2d, const float *conics, const float *colors,
    const float *opacities,
    const uint32_t *splat_strides, /* NULL: the reference's dense arrays (rows of 2 / 3 / channels / 1 floats); else 4 HOST
                                      ints: row strides in floats of means2d, conics, colors, opacities.  When all four
                                      are GS_ROW_FLOATS and the pointers are columns 0 / 2 / 6 / 5 of one 64-byte-aligned
                                      row buffer (channels <= 4), the kernels fetch whole splat rows; with more than 4
                                      channels the same holds for the geometry alone (means2d / conics / opacities at
                                      columns 0 / 2 / 5 with stride GS_ROW_FLOATS, colours in their own array) */
    const float *backgrounds, /* [C,channels] or NULL */
    const uint8_t *masks,     /* [C,tile_h,tile_w] or NULL */
    uint32_t image_width, uint32_t image_height, uint32_t tile_size,
    uint32_t tile_width, uint32_t tile_height,
    const int32_t *tile_offsets, const int32_t *flatten_ids,
    float *render_colors, /* [C,H,W,channels] */
    float *render_alphas, /* [C,H,W,1] */
    int32_t *last_ids,    /* [C,H,W] */
    const gs_raster_plan *plan, void *scratch, /* both NULL: no checkpoints (inference, or an unsegmented backward) */
    void *zero_fill, size_t zero_fill_bytes, /* optional side job (NULL, 0: none): a 16-byte aligned buffer this call
                                                zero-fills, its stores spread over the tile workgroups' last instructions --
                                                meant for the gradient rows the matching gs_rasterize_bwd accumulates into,
                                                which then need no fill pass of their own */
    gs_stream_t stream);

int32_t gs_rasterize_bwd(
    uint32_t C, uint32_t n_elems, uint32_t n_isects, uint32_t channels,
    const float *means2d, const float *conics, const float *colors,
    const float *opacities, const uint32_t *splat_strides /* as in gs_rasterize_fwd */,
    const float *backgrounds, const uint8_t *masks,
    uint32_t image_width, uint32_t image_height, uint32_t tile_size,
    uint32_t tile_width, uint32_t tile_height,
    const int32_t *tile_offsets, const int32_t *flatten_ids,
    const float *render_colors, /* forward output, or NULL (disables the segmented path) */
    const float *render_alphas, const int32_t *last_ids,
    const float *v_render_colors, const float *v_render_alphas /* or NULL (= zero) */,
    int64_t v_render_colors_pixel_stride,   /* element strides of v_render_colors: (channels, 1) for a dense */
    int64_t v_render_colors_channel_stride, /* [C,H,W,channels] tensor; (0, 0) for a broadcast scalar (the gradient
                                               of sum(render)): nothing has to be materialised */
    float *v_means2d_abs, /* [n_elems,2] or NULL */
    float *v_means2d,     /* [n_elems,2] */
    float *v_conics,      /* [n_elems,3] */
    float *v_colors,      /* [n_elems,channels] */
    float *v_opacities,   /* [n_elems] */
    int32_t packed16,     /* != 0: v_means2d is ONE zero-filled [n_elems,16] buffer receiving every gradient (the splat-row
                             columns, see above); v_conics / v_colors / v_opacities are ignored, v_means2d_abs only
                             selects absgrad (non-NULL).  One 64-byte row per splat lets the kernels add a whole splat's
                             gradient with a single L2 request.  1 requires channels <= 4.
                             2 (more than 4 channels): the GEOMETRY rows only -- v_means2d is the zero-filled [n_elems,16]
                             buffer (columns 0..5 and, with absgrad, 10 / 11), the colour gradients go to v_colors
                             [n_elems,channels] (zero-filled by the caller): the form the 5..32-channel kernels want,
                             and gs_projection_rows_bwd reads the rows in place */
    int64_t *det_accum,   /* NULL, or DETERMINISTIC mode (channels <= 4): a zero-filled int64 [n_elems,2,12] buffer.  The per-splat
                             sums are then accumulated in fixed point with integer atomics (the order in which the work
                             items reach a splat no longer matters; two accumulators per value: units of 2^-38 for
                             contributions below 2^10, of 2^-6 above) and converted into the float outputs by a second
                             kernel, which OVERWRITES every row of them: two runs give bit-identical gradients.  The
                             reference's float atomics are not reproducible either (rasterize_to_pixels_bwd.cu:243-274). */
    const gs_raster_plan *plan, void *scratch, /* the forward's plan and scratch (contents preserved), or NULL, NULL */
    gs_stream_t stream);

/* ------------------------------------------------------------------------
 * Q1 / Q2  per-splat quantize / dequantize straight-through estimators
 * replaces the torch ops of fake_quantize_ste / STE
 *   (gsplat/compression_simulation/ops.py:39-54, 57-75)
 * noise: out = clamp(x, lo, hi) + noise * q_step   (noise supplied by the caller so the
 *        RNG stream stays torch's), bwd: v_x = v_out where lo <= x <= hi else 0.
 * round: x <- clamp(x, lo, hi) IN PLACE (the reference mutates the parameter),
 *        out = round_half_even(((x - lo) / (hi - lo)) / q) * q * (hi - lo) + lo,
 *        q = 1/(2^bits - 1); bwd is the identity (no kernel).
 * All arithmetic is IEEE fp32 with no contraction, in the reference's op order.
 * activation (opt-in fusion, GS_ACT_*): out = act(quantized value) -- the torch.exp / torch.sigmoid the trainer applies to
 * the hooked log-scales / opacity logits (examples/simple_trainer.py:779-786) in the same pass; the backward multiplies by
 * act' (read off `out`, the forward's output).  GS_ACT_NONE is the reference's bit-exact path.
 * ---------------------------------------------------------------------- */
int32_t gs_quantize_noise_fwd(
    uint64_t n, const float *x, const float *noise,
    float lo, float hi, float q_step, int32_t activation, float *out, gs_stream_t stream);
int32_t gs_quantize_noise_bwd(
    uint64_t n, const float *x, const float *v_out,
    float lo, float hi, int32_t activation, const float *out /* forward output; NULL with GS_ACT_NONE */,
    float *v_x, gs_stream_t stream);
int32_t gs_quantize_round_fwd(
    uint64_t n, float *x_inplace, float lo, float hi, float range /* (float)(hi-lo) */,
    float q_step_norm /* (float)(1/(2^bits-1)) */, int32_t activation, float *out, gs_stream_t stream);
/* round mode with a fused activation: v_x = v_out * act'(out) for every element (without an activation the gradient is the
 * identity and there is nothing to launch) */
int32_t gs_quantize_round_bwd(
    uint64_t n, const float *v_out, int32_t activation, const float *out, float *v_x, gs_stream_t stream);

/* Multi-tensor "noise" quantizer with the noise generated in the kernel (the hooks of BASELINE config 3 in ONE launch each way;
 * the noise never exists in memory).  Bit-identical to  noise = torch.empty_like(x).uniform_(-0.5, 0.5)  followed by
 * gs_quantize_noise_fwd, tensor after tensor: the kernel evaluates the Philox4x32-10 draws torch's uniform_ kernel would have
 * made for (philox_seed, philox_offset) -- see csrc/quantize.hip for the counter scheme.  The caller reads seed and offset from
 * the device's default generator and advances it by gs_quantize_philox_advance(n, grid_cap) per tensor, so the process's RNG
 * stream is exactly the reference's.  grid_cap = CUs * (max threads per CU / 256) of the device (what torch caps uniform_'s grid
 * at).  descs: HOST array (copied into the kernel arguments). */
#define GS_QUANT_MULTI_MAX 8
typedef struct gs_quant_desc {
    uint64_t n;             /* floats in the tensor (0: skipped) */
    const float *x;         /* input */
    float *out;             /* fwd: output; bwd: the forward's output (only read with an activation) */
    const float *v_out;     /* bwd: upstream gradient */
    float *v_x;             /* bwd: gradient out */
    float lo, hi, q_step;
    int32_t activation;     /* GS_ACT_* fused behind the quantizer */
    uint64_t philox_offset; /* fwd: the generator's offset when this tensor's uniform_ call would have run (multiple of 4) */
} gs_quant_desc;
uint64_t gs_quantize_philox_advance(uint64_t n, uint32_t grid_cap);
int32_t gs_quantize_noise_multi_fwd(
    uint32_t n_tensors, const gs_quant_desc *descs, uint64_t philox_seed, uint32_t grid_cap, gs_stream_t stream);
int32_t gs_quantize_noise_multi_bwd(uint32_t n_tensors, const gs_quant_desc *descs, gs_stream_t stream);
/* The round mode for several tensors in one launch each way (round 6): gs_quantize_round_fwd's arithmetic per element.  descs[t].x is
 * the parameter and descs[t].v_x its WRITABLE alias (the same pointer: the parameter is clamped in place, ops.py:63), out the grid
 * value (activated when descs[t].activation is set); ranges[t] = hi - lo and q_step_norms[t] = 1 / (2^bits - 1), both computed as the
 * reference does (python double, then float); HOST arrays.  The backward is the identity (no call needed) except behind a fused
 * activation: gs_quantize_round_multi_bwd writes v_x = v_out x d act (out), no clamp mask, for the descriptors with n > 0. */
int32_t gs_quantize_round_multi_fwd(uint32_t n_tensors, const gs_quant_desc *descs, const float *ranges, const float *q_step_norms,
                                    gs_stream_t stream);
int32_t gs_quantize_round_multi_bwd(uint32_t n_tensors, const gs_quant_desc *descs, gs_stream_t stream);


/* ------------------------------------------------------------------------
 * Q3  learnable per-splat mask on the higher SH bands ("shN adaptive mask") of the compression-simulation hooks
 * replaces the torch ops of AnnealingMask (gsplat/compression_simulation/ada_mask.py:6-62), applied by
 * CompressionSimulation.simulate_compression_shN (simulation.py:319-324), and of the "gradient" strategy's
 * shN_gradient_threshold (simulation.py:327-348).
 *   x [n, row]: the shN parameter, row = 3 (K - 1) floats per splat (45 at SH degree 3); mask_logits [n].
 *   training (binary = 0): mask = sigmoid(logit / temperature); eval (binary = 1): mask = sigmoid(logit) >= 0.5 (ada_mask.py:39)
 *   fwd: out = x * mask.
 *   bwd: v_x = v_out * mask (NULL: not wanted); v_mask_logits[n] = (sum_j v_out[n,j] x[n,j]) * mask (1 - mask) / temperature
 *        (NULL: not wanted; must be NULL with binary = 1), every row written, reduced without atomics (deterministic).
 * IEEE fp32, no contraction, torch's operation order (true division by the temperature, sigmoid = 1 / (1 + exp(-v))).
 * ---------------------------------------------------------------------- */
int32_t gs_shn_mask_fwd(
    uint64_t n, uint32_t row, const float *x, const float *mask_logits, float temperature, int32_t binary,
    float *out, gs_stream_t stream);
int32_t gs_shn_mask_bwd(
    uint64_t n, uint32_t row, const float *x, const float *mask_logits, float temperature, int32_t binary,
    const float *v_out, float *v_x, float *v_mask_logits, gs_stream_t stream);
/* the mask itself, one float per splat (binary = 1: get_binary_mask, ada_mask.py:42-44) */
int32_t gs_mask_values(
    uint64_t n, const float *mask_logits, float temperature, int32_t binary, float *out, gs_stream_t stream);
/* out[0] = (sum_i mask_i) / divisor in fp32 -- the mean of the soft mask behind get_sparsity_loss (ada_mask.py:46-58:
 * divisor = n) or, with binary = 1, get_mask_ratio's count / shape[0] (ada_mask.py:60-62).  temp: gs_mask_sum_temp_bytes()
 * bytes of scratch; the sum is accumulated in double in a fixed order (deterministic). */
size_t gs_mask_sum_temp_bytes(void);
int32_t gs_mask_sum(
    uint64_t n, const float *mask_logits, float temperature, int32_t binary, float divisor, void *temp, float *out,
    gs_stream_t stream);
/* gradient of that mean (binary = 0): v_mask_logits[i] = (v_mean[0] / divisor) * (1 - y) y / temperature, y = sigmoid(logit_i / T) */
int32_t gs_mask_mean_bwd(
    uint64_t n, const float *mask_logits, float temperature, const float *v_mean /* device scalar */, float divisor,
    float *v_mask_logits, gs_stream_t stream);
/* "gradient" strategy (simulation.py:327-348), two launches, no host read-back: zero_rows[n] <- 1 where every value of x's
 * row is exactly 0, *n_zero <- their count; threshold = 2e-3 when 1 - n_zero / n < 0.10, else 100; the rows of grad_inplace
 * whose x row is all zero AND whose Frobenius norm is below the threshold are set to 0. */
int32_t gs_shn_grad_threshold(
    uint64_t n, uint32_t row, const float *x, float *grad_inplace, uint8_t *zero_rows /* [n] scratch / output */,
    uint64_t *n_zero /* device scalar, output */, gs_stream_t stream);

/* ------------------------------------------------------------------------
 * Factorized-prior bits estimator (the rate term behind the quantize hooks).
 * Replaces Entropy_factorized_optimized_refactor.forward
 * (gsplat/compression_simulation/entropy_model.py:195-254) and the autograd graph torch builds from
 * it: bits[n,c] = -log2(max(bound, |sigmoid(s u) - sigmoid(s l)|)), l/u = f_p(x[n,c] -/+ half_q[c]),
 * f_p a 1 -> W -> .. -> W -> 1 MLP (hidden_layers x hidden_width; reference: filters=(3,3) or (3,3,3))
 * with softplus'd matrices and tanh-gated residual nonlinearity (229-238), s = -sign(l+u) (247),
 * LowerBound gradient rule (355-357).  The parameter set of element (n,c) follows the reference's
 * 32-way reshape: p = (32 c + n / chunk) % channels, chunk = (n_rows + 32 - n_rows % 32) / 32.
 * params [channels, P] raw (un-transformed) parameters, per set and per layer [matrix row-major |
 * bias | factor], the last layer without factor; P = gs_entropy_factorized_params_per_channel().
 * x, bits, v_bits, v_x: [n_rows, channels] row-major.  half_q: [channels] device floats (= Q/2).
 * bwd: v_x is overwritten; v_params [replicas, channels, P] is ACCUMULATED with atomics (zero-fill first)
 * and the caller sums it over the leading axis -- workgroups spread over the replicas because
 * same-address device-scope float atomics serialise (replicas = 1 is valid, 32 is what the wrapper uses).
 * hidden_layers, hidden_width in 1..4 (uniform width), channels in 1..32; anything else: status 1. */
uint32_t gs_entropy_factorized_params_per_channel(uint32_t hidden_layers, uint32_t hidden_width);
int32_t gs_entropy_factorized_fwd(
    uint64_t n_rows, uint32_t channels, uint32_t hidden_layers, uint32_t hidden_width,
    const float *x, const float *half_q, const float *params, float likelihood_bound,
    float *bits, gs_stream_t stream);
int32_t gs_entropy_factorized_bwd(
    uint64_t n_rows, uint32_t channels, uint32_t hidden_layers, uint32_t hidden_width,
    const float *x, const float *half_q, const float *params, float likelihood_bound,
    const float *v_bits, float *v_x, float *v_params, uint32_t replicas, gs_stream_t stream);

/* ------------------------------------------------------------------------
 * Unfused public ops (reference: quat_scale_to_covar_preci_{fwd,bwd}.cu,
 * world_to_cam_{fwd,bwd}.cu, proj_{fwd,bwd}.cu)
 * ---------------------------------------------------------------------- */
int32_t gs_quat_scale_to_covar_preci_fwd(
    uint32_t N, const float *quats, const float *scales, int32_t triu,
    float *covars /* [N,3,3] or [N,6] or NULL */,
    float *precis /* same or NULL */, gs_stream_t stream);
int32_t gs_quat_scale_to_covar_preci_bwd(
    uint32_t N, const float *quats, const float *scales, int32_t triu,
    const float *v_covars, const float *v_precis, /* either may be NULL */
    float *v_quats, float *v_scales,              /* overwritten */
    gs_stream_t stream);

/* world_to_cam: means_c[c,n] = R_c p_n + t_c, covars_c[c,n] = R_c S_n R_c^T on GENERAL 3x3 matrices
 * (reference world_to_cam_fwd_tensor, csrc/world_to_cam_fwd.cu:85-127; device code
 * include/transform.cuh:8-46).  Either output (with its input) may be NULL.
 * bwd (world_to_cam_bwd_tensor, csrc/world_to_cam_bwd.cu:125-195; transform.cuh:19-68): v_means [N,3] and
 * v_covars [N,3,3] are OVERWRITTEN (summed over cameras inside the lane), v_viewmats [C,4,4] is
 * ACCUMULATED with atomics (zero-fill first); any of the three may be NULL (= not needed). */
int32_t gs_world_to_cam_fwd(
    uint32_t C, uint32_t N, const float *means /* [N,3] */, const float *covars /* [N,3,3] */,
    const float *viewmats /* [C,4,4] */, float *means_c /* [C,N,3] */, float *covars_c /* [C,N,3,3] */,
    gs_stream_t stream);
int32_t gs_world_to_cam_bwd(
    uint32_t C, uint32_t N, const float *means, const float *covars, const float *viewmats,
    const float *v_means_c /* [C,N,3] or NULL */, const float *v_covars_c /* [C,N,3,3] or NULL */,
    float *v_means, float *v_covars, float *v_viewmats, gs_stream_t stream);

/* proj: camera-space gaussians -> image plane, means2d [C,N,2], covars2d [C,N,2,2] = J S J^T with the
 * pinhole / orthographic / fisheye Jacobian (reference proj_fwd_tensor, csrc/proj_fwd.cu:81-129,
 * proj_bwd_tensor, csrc/proj_bwd.cu:128-182; device code include/proj.cuh).  No culling, general 3x3 S.
 * bwd overwrites v_means [C,N,3] and v_covars [C,N,3,3]. */
int32_t gs_proj_fwd(
    uint32_t C, uint32_t N, const float *means /* [C,N,3] */, const float *covars /* [C,N,3,3] */,
    const float *Ks /* [C,3,3] */, int32_t width, int32_t height, int32_t camera_model,
    float *means2d, float *covars2d, gs_stream_t stream);
int32_t gs_proj_bwd(
    uint32_t C, uint32_t N, const float *means, const float *covars, const float *Ks,
    int32_t width, int32_t height, int32_t camera_model,
    const float *v_means2d, const float *v_covars2d, float *v_means, float *v_covars, gs_stream_t stream);

/* rasterize_to_indices_in_range (reference rasterize_to_indices_in_range_tensor,
 * csrc/rasterize_to_indices_in_range.cu:177-300, kernel 16-175): for the list batches
 * [range_start, range_end) of every tile (one batch = tile_size^2 sorted entries) and the per-pixel
 * transmittances reached so far, list the (gaussian, pixel) pairs that get composited.
 * Two-call protocol: _count writes chunk_cnts i32 [C*H*W] (zero-fill first: untouched tiles stay 0), the
 * caller forms the exclusive prefix sum chunk_starts (i32, like the reference's int32 cumsum) and
 * the total, allocates, then _fill writes gaussian_ids (= flatten id % N) and pixel_ids
 * (= camera * H * W + row * W + column), both i64, pixel-major, list order inside a pixel. */
int32_t gs_rasterize_indices_count(
    uint32_t range_start, uint32_t range_end, uint32_t C, uint32_t N, uint32_t n_isects,
    const float *means2d, const float *conics, const float *opacities,
    uint32_t image_width, uint32_t image_height, uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    const int32_t *tile_offsets, const int32_t *flatten_ids, const float *transmittances,
    int32_t *chunk_cnts, gs_stream_t stream);
int32_t gs_rasterize_indices_fill(
    uint32_t range_start, uint32_t range_end, uint32_t C, uint32_t N, uint32_t n_isects,
    const float *means2d, const float *conics, const float *opacities,
    uint32_t image_width, uint32_t image_height, uint32_t tile_size, uint32_t tile_width, uint32_t tile_height,
    const int32_t *tile_offsets, const int32_t *flatten_ids, const float *transmittances,
    const int32_t *chunk_starts, int64_t *gaussian_ids, int64_t *pixel_ids, gs_stream_t stream);

/* ------------------------------------------------------------------------
 * On-disk attribute format: per-channel min-max quantization of a row-major
 * [rows, channels] grid and its exact inverse -- the arithmetic of _compress_png / _compress_png_kbit /
 * _compress_png_16bit and the matching _decompress_* (gsplat/compression/png_compression.py:166-389)
 * without the PNG container.  bits 1..8: ONE uint8 plane holding round(norm * (2^bits - 1)) << (8 - bits);
 * bits 16: low and high byte planes.  mins / maxs: [channels] device floats (torch.amin / amax of the grid).
 * Encode is fp32 with round-half-to-even, decode is float64 then cast, exactly like the numpy / torch mix
 * of the reference, so both are bit-exact against it. */
int32_t gs_grid_quantize(
    uint64_t n /* rows * channels */, uint32_t channels, const float *x, const float *mins, const float *maxs,
    uint32_t bits, uint8_t *plane_lo, uint8_t *plane_hi /* 16-bit only, else NULL */, gs_stream_t stream);
int32_t gs_grid_dequantize(
    uint64_t n, uint32_t channels, const uint8_t *plane_lo, const uint8_t *plane_hi /* or NULL */,
    const float *mins, const float *maxs, uint32_t bits, float *out, gs_stream_t stream);

/* Decode of the compressed attribute planes STRAIGHT INTO rasterization()'s inputs: replaces the
 * host-side chain of the reference's eval path -- PngCompression.decompress (gsplat/compression/png_compression.py:166-236:
 * _decompress_png_16bit for the means, _decompress_png_kbit / _decompress_png for scales, quats, opacities, sh0), the
 * inverse log transform of the means (228-230) and the trainer's activations (examples/simple_trainer.py:779-786) -- by one
 * kernel over the uint8 planes.  Planes are the [n, C] row-major views of the reference's [side, side, C] images; k-bit
 * planes keep their value in the top bits of the byte (as stored).  mins14 / maxs14 are HOST arrays: the per-channel
 * bounds of meta.json in the order means(3) scales(3) quats(4) opacities(1) sh0(3); bits5 (host): bit depth per attribute,
 * means = 16.  activate = 1: scales = exp(.), opacities = sigmoid(.) (what rasterization() takes); 0: the trainer's raw
 * parameters, bit-identical to gs_grid_dequantize.  normalize_quats = 1: quats / max(||quats||, 1e-12). */
int32_t gs_decode_splats(
    uint64_t n, const uint8_t *means_lo, const uint8_t *means_hi, const uint8_t *scales, const uint8_t *quats,
    const uint8_t *opacities, const uint8_t *sh0, const float *mins14, const float *maxs14, const uint32_t *bits5,
    int32_t normalize_quats, int32_t activate, float *means_out, float *scales_out, float *quats_out, float *opacities_out,
    float *sh0_out, gs_stream_t stream);
/* K-means codebook decode of the higher SH bands (png_compression.py:487-520): out[r, :] =
 * centroids_quant[labels[r], :] / (2^bits - 1) * (maxs - mins) + mins (float64 arithmetic like the reference).
 * labels are file contents: one outside [0, n_centroids) (where the reference's centroids[labels] raises IndexError)
 * decodes to a NaN row and is counted in *n_bad (device uint32, zero-filled by the caller; may be NULL). */
int32_t gs_kmeans_decode(uint64_t n_rows, uint32_t width, const int32_t *labels, const uint8_t *centroids_quant,
                         uint32_t n_centroids, uint32_t bits, float mins, float maxs, float *out, uint32_t *n_bad,
                         gs_stream_t stream);
/* PNG scanline reconstruction, HOST code (PNG specification section 9: filter types 0..4): data = h rows of 1 + stride bytes
 * (filter type, filtered scanline), out = h rows of stride bytes, bpp = bytes per pixel.  The sequential part of reading the
 * image grids the reference writes through imageio (gsplat/compression/png_compression.py:196, 271, 344-349). */
int32_t gs_png_unfilter(const uint8_t *data, uint32_t h, uint32_t stride, uint32_t bpp, uint8_t *out);


/* ------------------------------------------------------------------------
 * accumulate (reference gsplat/cuda/_torch_impl.py:432-519, exported as gsplat.accumulate): alpha compositing over an explicit list
 * of M (gaussian, pixel, camera) intersections -- the output of the rasterize_indices pair above: grouped by ray (camera, pixel),
 * front to back inside a ray.  alpha = min(opacity exp(-sigma), 0.999); weight = alpha x the product of (1 - alpha) over the
 * entries in front of it in its run (nerfacc.render_weight_from_alpha); renders[camera, pixel] += sum weight colour,
 * alphas[camera, pixel] += sum weight (nerfacc.accumulate_along_rays).  A run = consecutive entries with the same (camera, pixel).
 * renders [C,H,W,channels] / alphas [C,H,W] are ADDED to (the caller zero-fills them); alpha_buf / weights [M] are written and
 * handed back to gs_accumulate_bwd, which ADDS into v_means2d [C,N,2], v_conics [C,N,3], v_opacities [C,N], v_colors
 * [C,N,channels] (float atomics; each may be NULL) and uses v_alpha_pair [M] as scratch.  v_renders / v_alphas may be NULL. */
int32_t gs_accumulate_fwd(uint64_t M, uint32_t C, uint32_t N, uint32_t channels, const float *means2d, const float *conics,
                          const float *opacities, const float *colors, const int64_t *gaussian_ids, const int64_t *pixel_ids,
                          const int64_t *camera_ids, int32_t image_width, int32_t image_height, float *alpha_buf, float *weights,
                          float *renders, float *alphas, gs_stream_t stream);
int32_t gs_accumulate_bwd(uint64_t M, uint32_t C, uint32_t N, uint32_t channels, const float *means2d, const float *conics,
                          const float *opacities, const float *colors, const int64_t *gaussian_ids, const int64_t *pixel_ids,
                          const int64_t *camera_ids, int32_t image_width, int32_t image_height, const float *alpha_buf,
                          const float *weights, const float *v_renders, const float *v_alphas, float *v_alpha_pair,
                          float *v_means2d, float *v_conics, float *v_opacities, float *v_colors, gs_stream_t stream);

/* Accumulated depth -> expected depth, the image-level tail of rasterization()'s "ED" / "RGB+ED" render modes
 * (gsplat/rendering.py:471-477): out = cat(renders[..., :-1], renders[..., -1:] / alphas.clamp(min=1e-10)) over n_pix pixels of
 * `channels` floats, one pass each way instead of slice / clamp / div / cat and their autograd twins.  bwd: v_renders[..., :-1] =
 * v_out[..., :-1], v_renders[..., -1] = v_out[..., -1] / clamp(alpha), v_alphas = -v_out[..., -1] ((renders[..., -1] / clamp) / clamp)
 * where alpha >= 1e-10, else 0 (torch's div and clamp derivatives); either output may be NULL. */
int32_t gs_expected_depth_fwd(uint64_t n_pix, uint32_t channels, const float *renders, const float *alphas, float *out, gs_stream_t stream);
int32_t gs_expected_depth_bwd(uint64_t n_pix, uint32_t channels, const float *renders, const float *alphas, const float *v_out,
                              float *v_renders, float *v_alphas, gs_stream_t stream);

/* ------------------------------------------------------------------------
 * Temporal slicing of dynamic (spacetime) gaussians at one timestamp: the elementwise
 * chain in front of rasterization() in examples/simple_trainer_dyngs.py:506-521 -- trbf opacity decay
 * exp(-((t - center) / (sqrt2 scale))^2), cubic motion of the means (motion [N,9] = linear | quadratic |
 * cubic coefficients), quats + (t - center) omega re-normalised (F.normalize, eps 1e-12).  t - center is
 * detached where it drives motion and rotation (as `tforpoly` is), so trbf_center / trbf_scale receive
 * gradient through the opacity only.  trbf (optional output) is what the caller thresholds (> 0.05) for
 * the temporal visibility mask.  bwd: every output pointer may be NULL (not needed); all are overwritten. */
int32_t gs_temporal_slice_fwd(
    uint32_t n, const float *means, const float *motion, const float *quats, const float *omega,
    const float *opacities, const float *trbf_center, const float *trbf_scale, float timestamp,
    float *means_t, float *quats_t, float *opacity_t, float *trbf /* or NULL */, gs_stream_t stream);
int32_t gs_temporal_slice_bwd(
    uint32_t n, const float *means, const float *motion, const float *quats, const float *omega,
    const float *opacities, const float *trbf_center, const float *trbf_scale, float timestamp,
    const float *v_means_t, const float *v_quats_t, const float *v_opacity_t, const float *v_trbf /* each or NULL */,
    float *v_means, float *v_motion, float *v_quats, float *v_omega, float *v_opacities,
    float *v_trbf_center, float *v_trbf_scale, gs_stream_t stream);

/* The spacetime trainer's nine colour channels in one pass each way (round 6; examples/simple_trainer_STG.py:506-551):
 *     out[n] = (colors[n] | features_dir[n] | (timestamp - trbf_center[n]) * features_time[n]),   [N,9] from three [N,3]
 * replacing torch.cat + the tforpoly multiply (and autograd's split + copies on the way back).  quant_mask bit p (0 colors, 1
 * features_dir, 2 features_time): that part first goes through the round-to-grid STE hook of the compression simulation
 * (gsplat/compression_simulation/ops.py:57-75 with q_type "round": the PARAMETER is clamped in place to [lo, hi], the value used is
 * round((x - lo) / range / step_norm) * step_norm * range + lo with range = hi - lo, step_norm = 1 / (2^bits - 1), as
 * gs_quantize_round_fwd computes it); tables of 3 floats each, NULL with quant_mask 0.  bwd: the hook's gradient is the identity;
 * v_features_time = (timestamp - trbf_center) * v_out[:, 6:9]; trbf_center receives none (tforpoly is detached).  Output pointers of
 * bwd may be NULL (not needed). */
int32_t gs_stg_features_fwd(
    uint32_t n, float *colors, float *features_dir, float *features_time, const float *trbf_center, float timestamp,
    uint32_t quant_mask, const float *quant_lo, const float *quant_hi, const float *quant_range, const float *quant_step_norm,
    float *out, gs_stream_t stream);
int32_t gs_stg_features_bwd(
    uint32_t n, const float *v_out, const float *trbf_center, float timestamp, float *v_colors, float *v_features_dir,
    float *v_features_time, gs_stream_t stream);

/* The same slice evaluated INSIDE the row-form projection (round 6; SURVEY 8f rank 2: "folded into the projection kernel's load
 * phase"): gs_projection_rows_dyn_fwd == gs_temporal_slice_fwd followed by gs_projection_rows_fwd (colours [N,3] or none, no SH),
 * bit for bit -- the per-splat arithmetic is one shared definition (csrc/dynamic_dev.h, projection_dev.h) -- without the round trip of
 * means_t / quats_t / opacity_t through HBM; gs_projection_rows_dyn_bwd == gs_projection_rows_bwd followed by
 * gs_temporal_slice_bwd, writing the gradients of means, quats, scales, motion, omega, trbf_center, trbf_scale, opacities (and colors)
 * for the gaussians some camera saw (outputs_prefilled != 0: the other rows hold zeros already and are not stored; 0: every row is
 * written).  Opt-in on top, so that the trainer's RAW parameters can be handed over as they are (simple_trainer_dyngs.py:493-505):
 *   raw_params   GS_DYN_RAW_* bits: scales are log-scales (exp), opacities logits (sigmoid), trbf_scale a log-scale (exp) -- the
 *                activation runs in the kernel, its derivative in the backward;
 *   quant_mask   bit 0 scales, 1 quats, 2 opacities, 3 colors: the attribute goes through the STE round quantizer first
 *                (gsplat/compression_simulation/ops.py:57-75, the arithmetic of gs_quantize_round_fwd bit for bit): clamp to
 *                [quant_lo[k], quant_hi[k]] -- stored back into the PARAMETER when it changes a value, as the reference's in-place
 *                clamp does, which is why quats / scales / opacities / colors are not const in the forward --, round to the grid
 *                (quant_range[k] = hi - lo, quant_step_norm[k] = 1 / (2^bits - 1), both computed as the reference does: python
 *                double, then float), identity gradient.  The four tables are HOST arrays of 4 floats (NULL with quant_mask 0).
 * colors NULL: the colour columns of the rows are left to the caller (more than three render channels). */
#define GS_DYN_RAW_SCALES 1u
#define GS_DYN_RAW_OPACITIES 2u
#define GS_DYN_RAW_TRBF_SCALE 4u
int32_t gs_projection_rows_dyn_fwd(
    uint32_t C, uint32_t N, const float *means, float *quats, float *scales, const float *motion, const float *omega,
    const float *trbf_center, const float *trbf_scale, float timestamp,
    float min_trbf /* gaussians whose temporal basis is <= this are culled at this timestamp (the trainer's temp_vis_mask, 0.05); < 0: none */,
    uint8_t *trbf_alive /* [N] or NULL: 1 where trbf > min_trbf (the trainer's t_vis_mask) */,
    uint32_t raw_params, uint32_t quant_mask, const float *quant_lo,
    const float *quant_hi, const float *quant_range, const float *quant_step_norm, const float *viewmats, const float *Ks,
    int32_t image_width, int32_t image_height, float eps2d, float near_plane, float far_plane, float radius_clip, int32_t camera_model,
    float *opacities /* [N] or NULL */, float *colors /* [N,3] or NULL */, int32_t antialiased,
    uint32_t tile_size, uint32_t tile_width, uint32_t tile_height, int32_t *tiles_per_gauss /* or NULL */, int32_t *block_sums /* or NULL */,
    int32_t *radii, float *depths, float *rows, gs_stream_t stream);
int32_t gs_projection_rows_dyn_bwd(
    uint32_t C, uint32_t N, const float *means, const float *quats, const float *scales, const float *motion, const float *omega,
    const float *trbf_center, const float *trbf_scale, float timestamp, uint32_t raw_params, uint32_t quant_mask, const float *quant_lo,
    const float *quant_hi, const float *quant_range, const float *quant_step_norm, const float *viewmats, const float *Ks,
    int32_t image_width, int32_t image_height, float eps2d, int32_t camera_model, const int32_t *radii, const float *rows,
    const float *grad_rows, const float *v_depths /* or NULL */, const float *opacities, int32_t antialiased,
    float *v_means, float *v_quats, float *v_scales, float *v_motion, float *v_omega, float *v_trbf_center, float *v_trbf_scale,
    float *v_opacities, float *v_colors /* each or NULL */, int32_t outputs_prefilled, gs_stream_t stream);

/* ------------------------------------------------------------------------
 * Row packing around the multi-GPU exchange of projected splats.  The gaussian-sharded mode of the reference
 * (gsplat/rendering.py:397-478, gsplat/distributed.py:170-257) concatenates radii | means2d | depths | conics |
 * opacities | colours with torch.cat before every all-to-all and splits them afterwards; these two replace the
 * cat / split: gs_rows_pack gathers the column blocks of up to 8 row-major arrays of 4-byte elements (part k:
 * device pointer parts[k] or NULL = zeros, widths[k] columns, row stride row_strides[k] elements, so column views
 * of wider buffers are read in place) into wire rows of sum(widths) elements; gs_rows_unpack scatters wire rows
 * back (NULL part = skipped).  parts / widths / row_strides are HOST arrays of n_parts entries. */
int32_t gs_rows_pack(
    uint64_t n_rows, int32_t n_parts, const void *const *parts, const int32_t *widths, const int64_t *row_strides,
    void *wire, gs_stream_t stream);
int32_t gs_rows_unpack(
    uint64_t n_rows, int32_t n_parts, void *const *parts, const int32_t *widths, const int64_t *row_strides,
    const void *wire, gs_stream_t stream);
/* Sparse form (only the rows of visible splats travel): wire row r pairs with row row_index[r * index_stride] of every
 * part whose indexed[k] flag (HOST array) is non-zero, and with row r of the others -- a gather on the way in, a scatter
 * on the way out.  row_index may alias a column of the wire itself (index_stride = wire width). */
int32_t gs_rows_pack_indexed(
    uint64_t n_rows, int32_t n_parts, const void *const *parts, const int32_t *widths, const int64_t *row_strides,
    const int32_t *indexed, const int32_t *row_index, int64_t index_stride, void *wire, gs_stream_t stream);
int32_t gs_rows_unpack_indexed(
    uint64_t n_rows, int32_t n_parts, void *const *parts, const int32_t *widths, const int64_t *row_strides,
    const int32_t *indexed, const int32_t *row_index, int64_t index_stride, const void *wire, gs_stream_t stream);
/* Compaction for the sparse form of that exchange: the rows with radii > 0 of radii [C_total, N] are listed per
 * destination rank d = c / C_local in a chunk of `cap` slots + 1 header row (slot order arbitrary):
 * src_index [world * (cap + 1)] = c * N + n or -1; hdr [world * (cap + 1), 2] = (row in the receiver's [C_local * N_total]
 * arrays, 0) or -1, header row = (-1, min(count, cap) | overflow << 30) where overflow = SOME chunk of this call was too small; counters [world] = rows wanted per destination;
 * stats [2] = (largest count, any overflow).  Both index arrays feed gs_rows_pack_indexed (a negative index packs
 * zeros / is skipped by gs_rows_unpack_indexed).  No read-back: `cap` comes from the caller's previous steps, and an
 * overflow is visible to every receiver in the header rows. */
int32_t gs_exchange_compact(
    uint32_t C_total, uint32_t N, uint32_t C_local, uint32_t world, uint32_t cap, uint32_t N_total, uint32_t N_off,
    const int32_t *radii, int32_t *src_index, int32_t *hdr, uint32_t *counters, uint32_t *stats, gs_stream_t stream);
/* The same exchange for SPLAT ROWS (gs_projection_rows_fwd): the 64-byte row IS the wire row, so nothing is packed or
 * split -- a row carries its radius (column 10) and depth (column 9), and columns 12 / 13 (padding) carry the two ints of
 * gs_exchange_compact's `hdr` (destination row | chunk header).
 * gs_rows16_gather: out_rows[r] = src_rows[index[r * index_stride]] (a negative index gives zeros); tag [n_rows,2] or NULL
 *   is stored into columns 12 / 13.  Forward: index = src_index, tag = hdr; backward: index = column 12 of the received
 *   rows (index_stride 16), src_rows = the gradient rows of gs_rasterize_bwd, tag = NULL.
 * gs_rows16_scatter: dst_rows[index[r * index_stride]] = wire_rows[r] where the index is >= 0; radii / depths (optional)
 *   receive columns 10 / 9 at the same element.  Forward: index = column 12 of the received rows; backward: src_index. */
int32_t gs_rows16_gather(
    uint64_t n_rows, const int32_t *index, int64_t index_stride, const float *src_rows, const int32_t *tag, float *out_rows,
    gs_stream_t stream);
int32_t gs_rows16_scatter(
    uint64_t n_rows, const int32_t *index, int64_t index_stride, const float *wire_rows, float *dst_rows, int32_t *radii,
    float *depths, gs_stream_t stream);
/* The two sides of that exchange as ONE call each (what distributed._ExchangeRows issues: the gaussian-sharded step is bound
 * by the host's launch work, every native call less counts):
 * gs_exchange_rows_send = gs_exchange_compact + gs_rows16_gather(src_index, hdr) into send_rows [world * (cap + 1), 16];
 *   src_index is kept for the backward, hdr / counters are scratch of the call.
 *   zero_radii (optional, [n_zero] int32): zero-filled by the call's first launch -- the RECEIVER-side radii of the same
 *   process, allocated before the all-to-all; gs_exchange_rows_recv(radii_zeroed = 1) then skips its own fill.
 * gs_exchange_rows_recv = zero-fill of radii [n_dst] + gs_rows16_scatter(index = column 12 of recv_rows) + gs_exchange_flags. */
int32_t gs_exchange_rows_send(
    uint32_t C_total, uint32_t N, uint32_t C_local, uint32_t world, uint32_t cap, uint32_t N_total, uint32_t N_off,
    const int32_t *radii, const float *rows, int32_t *src_index, int32_t *hdr, uint32_t *counters, uint32_t *stats,
    float *send_rows, int32_t *zero_radii, uint64_t n_zero, gs_stream_t stream);
int32_t gs_exchange_rows_recv(
    uint64_t n_recv, const float *recv_rows, uint64_t n_dst, float *dst_rows, int32_t *radii, float *depths,
    uint32_t world, const int64_t *hdr_rows, const uint32_t *stats, int32_t *out3, int32_t radii_zeroed, gs_stream_t stream);
/* After the all-to-all of those chunks: out3 = (some sender overflowed (bit 30 of the count in the header row hdr_rows[d] of
 * every received chunk; recv rows are row_width ints wide), stats[0], stats[1]).  out3 may be pinned HOST memory: the flags
 * then reach the host with the renderer's own tile-count read-back, without a copy command or a sync of their own. */
int32_t gs_exchange_flags(
    uint32_t world, const int32_t *recv, uint32_t row_width, const int64_t *hdr_rows, const uint32_t *stats, int32_t *out3,
    gs_stream_t stream);
/* ------------------------------------------------------------------------
 * Row gather and its adjoint for the packed (COO) pipeline: the torch indexing `opacities[gaussian_ids]`,
 * `colors[gaussian_ids]`, `means[gaussian_ids]` of gsplat/rendering.py:325, 365-380 and its backward (torch sorts the
 * ids and runs ~45 small kernels there).  src [*, width] fp32, ids int64 [n_rows]; gs_scatter_add_rows_f32 ADDS
 * v_out [n_rows, width] into v_src [*, width] (zero-fill first) with float atomics. */
int32_t gs_gather_rows_f32(
    uint64_t n_rows, uint32_t width, const float *src, const int64_t *ids, float *out, gs_stream_t stream);
int32_t gs_scatter_add_rows_f32(
    uint64_t n_rows, uint32_t width, const float *v_out, const int64_t *ids, float *v_src, gs_stream_t stream);
/* Plan of the sparse gradient reduction of the camera-sharded mode (distributed.py; no reference counterpart: the reference's
 * multi-GPU mode shards the gaussians).  gs_dp_visibility: vis[n] = any camera c with radii[c, n] > 0 (zeros behind N, up to
 * n_pad).  gs_dp_plan, from the all-gathered masks uint8 [world, n_pad = world * block] (splat n belongs to owner n / block):
 *   counts   int32 [world * world + world], ZERO-FILLED by the caller: rows[r][o] = splats of owner o that rank r saw, then
 *            urows[o] = splats of owner o that any rank saw;
 *   send_idx int32 [n_pad]: this rank's visible splats, ascending; urank int32 [n_pad]: position of every union splat in
 *            the ascending list of all union splats; uidx int32 [n_pad]: that list (entries behind the counts: undefined);
 *   tile_counts: scratch, 8 bytes x gs_dp_plan_tiles(n_pad).  world <= 16.
 * gs_dp_reduce_rows: the owner side of the reduction.  wire [n_recv, 1 + width] = the rows received from all senders (chunk
 * of sender k = rows [chunk_starts[k], chunk_starts[k + 1]), HOST array of world + 1 entries), column 0 the global splat index
 * as an int32 bit pattern (negative: no row).  Writes acc [umax, 1 + width]: row u = (uidx[u] for u < n_valid else -1 | scale *
 * sum of the received rows whose index maps to u, u = map[index] - map_offset) -- every row written once, no atomics, no
 * zero-fill.  inv: scratch, int32 [world * umax]. */
uint32_t gs_dp_plan_tiles(uint32_t n_pad);
int32_t gs_dp_visibility(uint32_t C, uint32_t N, uint32_t n_pad, const int32_t *radii, uint8_t *vis, gs_stream_t stream);
int32_t gs_dp_plan(uint32_t world, uint32_t rank, uint32_t n_pad, uint32_t block, const uint8_t *masks, void *tile_counts,
                   int32_t *counts, int32_t *send_idx, int32_t *urank, int32_t *uidx, gs_stream_t stream);
int32_t gs_dp_reduce_rows(uint64_t n_recv, uint32_t width, uint32_t world, const float *wire, const int64_t *chunk_starts,
                          const int32_t *map, int32_t map_offset, uint64_t umax, uint32_t n_valid, const int32_t *uidx, float scale,
                          int32_t *inv, float *acc, gs_stream_t stream);

/* ------------------------------------------------------------------------
 * Native step driver (round 4): the launches of one rasterization() forward + backward of the common training case --
 * unpacked batch, quats + scales or covars, shared SH coefficients (contiguous or split) or [N,3] colours, three render
 * channels, fixed camera poses -- issued from ONE descriptor in three calls instead of nine operator calls from the host
 * language.  Every launch goes through the operator entry points above: identical results; the operators stay the drop-in
 * boundary, this is the executor around them (the counterpart of gsplat/rendering.py:28-582's orchestration).
 *   gs_step_fwd_begin   gs_projection_rows_fwd (gs_projection_rows_dyn_fwd with dyn_motion) -> [gs_presort_split] -> gs_isect_count_keys -> gs_presort_buckets |
 *                       gs_sort_pairs_u64_i32_drop [-> gs_cumsum_i32 when group_prefix is given]
 *   (the caller waits until every entry of block_sums -- pinned host memory it pre-set to -1 -- is >= 0, sets n_isects /
 *    n_kept_host to the sums of its even / odd entries, sizes the phase-2 buffers with gs_isect_finish_work_bytes /
 *    gs_rasterize_plan and allocates them)
 *   gs_step_fwd_finish  gs_isect_finish_presorted -> gs_rasterize_fwd (zero-filling zero_fill as its side job)
 *   gs_step_bwd         gs_rasterize_bwd (packed gradient rows) -> gs_projection_rows_bwd
 * With rows_ready the first call starts at gs_isect_count_keys (binning + compositing of rows some other producer wrote).
 * All pointers are device pointers except block_sums (pinned host).  Buffers: radii i32 [C,N], depths [C,N], rows [C,N,16]
 * (64-byte aligned), tiles_per_gauss i32 [C,N], depth_keys i64 [C N], depth_vals i32 [C N] (radix pre-sort only), sort_temp
 * (gs_presort_temp_bytes / gs_sort_temp_bytes of C N), splitters i64 [256] (bucketed), sorted_keys i64 [C N] (radix), perm
 * i32 [C N], n_kept u32 [1], group_sums u32 [ceil(C N / 2^gs_isect_emit_group_shift())], group_prefix i64 of the same length
 * + cumsum_scratch (both or neither), block_sums i32 [C * gs_projection_rows_blocks(N)][2] ([gs_isect_count_blocks(C N)][2] with rows_ready); isect_ids i64 / flatten_ids i32
 * [n_isects], offsets i32 [C, tile_height, tile_width], work (gs_isect_finish_work_bytes), render_colors [C,H,W,3],
 * render_alphas [C,H,W,1], last_ids i32 [C,H,W], scratch (plan.scratch_bytes; NULL: no checkpoints), zero_fill: the gradient rows
 * [C N,16] (+ whatever else the caller wants zeroed behind them); backward: grad_rows = that zero-filled buffer, v_* outputs as
 * in gs_projection_rows_bwd (NULL: not wanted). */
typedef struct gs_step {
    uint32_t C, N;
    const float *means, *covars, *quats, *scales, *viewmats, *Ks, *opacities, *colors, *sh_coeffs, *sh_rest;
    uint32_t sh_K, sh_degree;
    int32_t width, height;
    float eps2d, near_plane, far_plane, radius_clip;
    int32_t camera_model, antialiased;
    uint32_t tile_size, tile_width, tile_height;
    int32_t bucketed; /* != 0: the bucketed depth pre-sort where gs_presort_applicable(C N) */
    uint32_t lds_capacity;
    int32_t sh_mask_binary;
    const float *sh_mask_logits; /* the shN mask fused into the SH evaluation (split rows), or NULL */
    float *v_sh_mask_logits;     /* backward: [N] or NULL */
    float sh_mask_temperature;
    uint32_t rows_ready; /* != 0: rows / radii / depths are inputs (e.g. received through the gaussian-sharded exchange): no projection,
                          * gs_isect_count_keys counts the tiles; block_sums then has gs_isect_count_blocks(C N) entries */
    const float *backgrounds; /* [C,3] or NULL */
    /* phase 1 */
    int32_t *radii;
    float *depths, *rows;
    int32_t *tiles_per_gauss;
    int64_t *depth_keys;
    int32_t *depth_vals;
    void *sort_temp;
    uint64_t sort_temp_bytes;
    int64_t *splitters, *sorted_keys;
    int32_t *perm;
    uint32_t *n_kept, *group_sums;
    int64_t *group_prefix;
    void *cumsum_scratch;
    uint64_t cumsum_scratch_bytes;
    int32_t *block_sums;
    /* phase 2 */
    uint64_t n_isects;
    uint32_t n_kept_host; /* elements the pre-sort kept (sum of the odd entries of block_sums); 0: unknown (no packed pairs) */
    uint32_t reserved1;
    int64_t *isect_ids;
    int32_t *flatten_ids, *offsets;
    void *work;
    uint64_t work_bytes;
    float *render_colors, *render_alphas;
    int32_t *last_ids;
    gs_raster_plan plan;
    void *scratch, *zero_fill;
    uint64_t zero_fill_bytes;
    /* backward */
    const float *v_render_colors, *v_render_alphas;
    int64_t vrc_pixel_stride, vrc_channel_stride;
    float *grad_rows;
    const float *v_depths;
    float *v_means, *v_covars, *v_quats, *v_scales, *v_opacities, *v_colors, *v_sh, *v_sh_rest;
    int32_t absgrad, outputs_prefilled, skip_projection_bwd;
    int32_t finish_phase; /* gs_step_fwd_finish: 0 = binning + compositing, 1 = binning only, 2 = compositing only */
    /* dynamic splats (round 6): dyn_motion != NULL routes the projection through gs_projection_rows_dyn_fwd / _bwd (no SH, no covars;
     * quats / scales / opacities / colors are then written when dyn_quant_mask clamps a parameter) */
    const float *dyn_motion, *dyn_omega, *dyn_trbf_center, *dyn_trbf_scale;
    float dyn_timestamp;
    uint32_t dyn_raw_params, dyn_quant_mask;
    float dyn_min_trbf; /* < 0: no temporal culling */
    float dyn_quant_lo[4], dyn_quant_hi[4], dyn_quant_range[4], dyn_quant_step_norm[4];
    float *v_dyn_motion, *v_dyn_omega, *v_dyn_trbf_center, *v_dyn_trbf_scale;
    uint8_t *dyn_trbf_alive; /* [N] or NULL */
} gs_step;
/* Layout guard for bindings that mirror the host structs by hand (ctypes, cgo, JNA ...): writes up to n entries --
 * sizeof(struct), then offsetof of the listed fields in this order -- and returns how many the list has.
 *   gs_step:       C, sh_K, eps2d, tile_size, sh_mask_logits, rows_ready, backgrounds, radii, sort_temp_bytes, block_sums, n_isects,
 *                  n_kept_host, work_bytes, plan, scratch, zero_fill_bytes, v_render_colors, vrc_pixel_stride, grad_rows, v_sh_rest, absgrad,
 *                  finish_phase, dyn_motion, dyn_timestamp, dyn_quant_lo, v_dyn_motion
 *   gs_quant_desc: n, x, out, v_out, v_x, lo, q_step, activation, philox_offset */
uint32_t gs_step_layout(uint64_t *out, uint32_t n);
uint32_t gs_quant_desc_layout(uint64_t *out, uint32_t n);
int32_t gs_step_fwd_begin(gs_step *step, gs_stream_t stream);
int32_t gs_step_fwd_finish(gs_step *step, gs_stream_t stream);
int32_t gs_step_bwd(gs_step *step, gs_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GSPLAT_HIP_H */
