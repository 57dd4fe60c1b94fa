"""Fast path of ``rasterization()`` over the native step driver (``gs_step_fwd_begin`` / ``gs_step_fwd_finish``, csrc/step.hip).

The common training call -- unpacked batch, shared SH coefficients (contiguous, or the trainer's ``(sh0, shN)`` pair) or
``[N, 3]`` colours, three render channels, fixed camera poses -- runs the reference's pipeline
(gsplat/rendering.py:279-582: projection, tile binning, compositing) as TWO native calls around the one host read-back
instead of seven operator calls with their Python glue: ~0.6 ms of host work per forward + backward becomes ~0.3 ms
(tools/cpu_overhead.py), with the same launches, i.e. identical results (tests/test_gpu_step.py).

Autograd sees the same two nodes as on the operator path, so that ``meta["means2d"]`` stays an autograd intermediate
(``retain_grad()`` / ``.absgrad`` of the reference's densification strategies, strategy/default.py:150, 221-226):

* ``_StepProject``   forward: the WHOLE forward (both native calls); backward: ``_ProjectRows.backward`` (projection + SH);
* ``_StepComposite`` forward: hands out the images node 1 rendered; backward: ``_RasterizeToPixels.backward``.

The backward bodies are the operator path's own (same saved tensors, same context attributes), so every property tested
there -- prefilled gradients, repeated backward, partial requires_grad, expanded image gradients -- carries over.
"""
from __future__ import annotations

import ctypes
import math
import os
from typing import Optional

import torch
from torch import Tensor

from . import _backend as B
from . import _wrapper as W

ENABLED = os.environ.get("GS_STEP_DRIVER", "1") != "0"


class _Plan(ctypes.Structure):  # gs_raster_plan
    _fields_ = [("magic", ctypes.c_uint32), ("n_tiles_all", ctypes.c_uint32), ("n_isects", ctypes.c_uint32), ("channels", ctypes.c_uint32),
                ("seg", ctypes.c_int32), ("solo_min", ctypes.c_int32), ("xcd_fwd", ctypes.c_uint32), ("xcd_bwd", ctypes.c_uint32),
                ("scratch_bytes", ctypes.c_uint64), ("reserved", ctypes.c_uint32 * 6)]


_P, _U32, _I32, _U64, _I64, _F = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_int32, ctypes.c_uint64, ctypes.c_int64, ctypes.c_float


class _Step(ctypes.Structure):  # gs_step of include/gsplat_hip.h (field for field)
    _fields_ = [
        ("C", _U32), ("N", _U32),
        ("means", _P), ("covars", _P), ("quats", _P), ("scales", _P), ("viewmats", _P), ("Ks", _P), ("opacities", _P), ("colors", _P),
        ("sh_coeffs", _P), ("sh_rest", _P),
        ("sh_K", _U32), ("sh_degree", _U32), ("width", _I32), ("height", _I32),
        ("eps2d", _F), ("near_plane", _F), ("far_plane", _F), ("radius_clip", _F),
        ("camera_model", _I32), ("antialiased", _I32), ("tile_size", _U32), ("tile_width", _U32), ("tile_height", _U32),
        ("bucketed", _I32), ("lds_capacity", _U32), ("sh_mask_binary", _I32),
        ("sh_mask_logits", _P), ("v_sh_mask_logits", _P), ("sh_mask_temperature", _F), ("rows_ready", _U32),
        ("backgrounds", _P),
        ("radii", _P), ("depths", _P), ("rows", _P), ("tiles_per_gauss", _P), ("depth_keys", _P), ("depth_vals", _P),
        ("sort_temp", _P), ("sort_temp_bytes", _U64), ("splitters", _P), ("sorted_keys", _P), ("perm", _P), ("n_kept", _P),
        ("group_sums", _P), ("group_prefix", _P), ("cumsum_scratch", _P), ("cumsum_scratch_bytes", _U64), ("block_sums", _P),
        ("n_isects", _U64), ("n_kept_host", _U32), ("reserved1", _U32), ("isect_ids", _P), ("flatten_ids", _P), ("offsets", _P), ("work", _P), ("work_bytes", _U64),
        ("render_colors", _P), ("render_alphas", _P), ("last_ids", _P), ("plan", _Plan), ("scratch", _P), ("zero_fill", _P),
        ("zero_fill_bytes", _U64),
        ("v_render_colors", _P), ("v_render_alphas", _P), ("vrc_pixel_stride", _I64), ("vrc_channel_stride", _I64),
        ("grad_rows", _P), ("v_depths", _P), ("v_means", _P), ("v_covars", _P), ("v_quats", _P), ("v_scales", _P),
        ("v_opacities", _P), ("v_colors", _P), ("v_sh", _P), ("v_sh_rest", _P),
        ("absgrad", _I32), ("outputs_prefilled", _I32), ("skip_projection_bwd", _I32), ("finish_phase", _I32),
        ("dyn_motion", _P), ("dyn_omega", _P), ("dyn_trbf_center", _P), ("dyn_trbf_scale", _P),
        ("dyn_timestamp", _F), ("dyn_raw_params", _U32), ("dyn_quant_mask", _U32), ("dyn_min_trbf", _F),
        ("dyn_quant_lo", _F * 4), ("dyn_quant_hi", _F * 4), ("dyn_quant_range", _F * 4), ("dyn_quant_step_norm", _F * 4),
        ("v_dyn_motion", _P), ("v_dyn_omega", _P), ("v_dyn_trbf_center", _P), ("v_dyn_trbf_scale", _P), ("dyn_trbf_alive", _P),
    ]


_LAYOUT_FIELDS = ("C", "sh_K", "eps2d", "tile_size", "sh_mask_logits", "rows_ready", "backgrounds", "radii", "sort_temp_bytes", "block_sums",
                  "n_isects", "n_kept_host", "work_bytes", "plan", "scratch", "zero_fill_bytes", "v_render_colors", "vrc_pixel_stride", "grad_rows",
                  "v_sh_rest", "absgrad", "finish_phase", "dyn_motion", "dyn_timestamp", "dyn_quant_lo", "v_dyn_motion")


def check_layout() -> None:
    """The ctypes mirror above against the library's own ``sizeof`` / ``offsetof`` of ``gs_step`` (``gs_step_layout``)."""
    want = (ctypes.c_uint64 * 64)()
    m = int(B.query("gs_step_layout", want, 64))
    mine = [ctypes.sizeof(_Step)] + [getattr(_Step, f).offset for f in _LAYOUT_FIELDS]
    if m != len(mine) or list(want[:m]) != mine:
        raise ImportError(f"gs_step: the ctypes mirror in _step.py does not match the library's struct layout ({list(want[:m])} vs {mine})")


_ROW_STRIDES = (ctypes.c_uint32 * 4)(W.ROW, W.ROW, W.ROW, W.ROW)


class _Handover:
    """What node 1's forward leaves for node 2 (the rendered images and everything its backward saves)."""

    __slots__ = ("render_colors", "render_alphas", "last_ids", "scratch", "plan", "grad_rows", "offsets", "flatten_ids")


def applicable(means: Tensor, viewmats: Tensor, colors, sh_degree, packed: bool, distributed: bool, render_mode: str,
               channel_chunk: int, deterministic: bool, fuse_sh: bool, row_colors) -> bool:
    return (ENABLED and not packed and not distributed and not deterministic and render_mode == "RGB" and channel_chunk >= 3
            and means.is_cuda and viewmats.is_cuda and not viewmats.requires_grad and means.shape[0] > 0 and viewmats.shape[0] > 0
            and (fuse_sh or row_colors is not None))


def _c(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32 tensor, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _phase1(s: "_Step", C: int, N: int, dev, n_sums: int, given=None) -> dict:
    """Everything whose size follows from C * N: allocates the phase-1 buffers of ``gs_step`` and enters them into ``s``.
    ``given`` = (radii, depths, rows) when they are inputs (rows_ready)."""
    n_elems = C * N
    i32, i64, f32, u8 = torch.int32, torch.int64, torch.float32, torch.uint8
    empty = torch.empty
    ptr = B.ptr
    if given is None:
        radii = empty((C, N), dtype=i32, device=dev)
        depths = empty((C, N), dtype=f32, device=dev)
        rows = empty((C, N, W.ROW), dtype=f32, device=dev)
    else:
        radii, depths, rows = given
    tiles_per_gauss = empty((C, N), dtype=i32, device=dev)
    dkeys = empty(n_elems, dtype=i64, device=dev)
    perm = empty(n_elems, dtype=i32, device=dev)
    n_kept = empty(1, dtype=i32, device=dev)
    gshift = _GSHIFT[0] or _init_consts()
    n_groups = (n_elems + (1 << gshift) - 1) >> gshift
    gsums = empty(n_groups, dtype=i32, device=dev)
    bucketed = W._PRESORT["on"] and bool(B.query("gs_presort_applicable", n_elems))
    dvals = None if bucketed else empty(n_elems, dtype=i32, device=dev)  # (bucketed: the element rides in the key's low half)
    tb = B.query("gs_presort_temp_bytes" if bucketed else "gs_sort_temp_bytes", n_elems)
    temp = empty(tb, dtype=u8, device=dev)
    split = W.presort_split_buffer(dev) if bucketed else None
    ko = None if bucketed else empty(n_elems, dtype=i64, device=dev)
    gpre = scratch1 = None
    if n_groups > _PREFIX_FROM[0]:
        gpre = empty(n_groups, dtype=i64, device=dev)
        sb1 = B.query("gs_cumsum_scratch_bytes", n_groups)
        scratch1 = empty(sb1, dtype=u8, device=dev)
        s.cumsum_scratch, s.cumsum_scratch_bytes = ptr(scratch1), sb1
    pinned = W._pinned_take(2 * n_sums)  # [n_sums][2]: (intersections, visible elements) per block
    s.radii, s.depths, s.rows, s.tiles_per_gauss = ptr(radii), ptr(depths), ptr(rows), ptr(tiles_per_gauss)
    s.depth_keys, s.depth_vals, s.sort_temp, s.sort_temp_bytes = ptr(dkeys), ptr(dvals), ptr(temp), tb
    s.splitters, s.sorted_keys, s.perm, s.n_kept, s.group_sums, s.group_prefix = ptr(split), ptr(ko), ptr(perm), ptr(n_kept), ptr(gsums), ptr(gpre)
    s.block_sums = pinned.data_ptr()
    s.bucketed, s.lds_capacity = int(bucketed), W._PRESORT["lds_capacity"]
    # (the dict keeps every buffer alive until the forward's launches are queued)
    return {"radii": radii, "depths": depths, "rows": rows, "tiles_per_gauss": tiles_per_gauss, "pinned": pinned,
            "keep": (dkeys, dvals, perm, n_kept, gsums, temp, split, ko, gpre, scratch1)}


_LEAKED: list = []  # pinned buffers a kernel may still write to (never handed back: see _drop_pinned)


def _drop_pinned(bufs: Optional[dict], n_sums: int) -> None:
    """Give the pinned block-sum buffer of an abandoned forward back.  The count kernel stores into it from the GPU and torch's
    pinned allocator does not see kernel stores: the buffer may only be reused once every sum has landed -- bounded wait, and
    if the kernel never gets there (a failed launch in front of it) the buffer is kept alive for good instead."""
    if not bufs or bufs.get("pinned") is None:
        return
    import time

    pinned, bufs["pinned"] = bufs["pinned"], None
    ev = W._SentinelEvent(pinned)
    deadline = time.perf_counter() + 1.0
    while not ev.query():
        if time.perf_counter() > deadline:
            _LEAKED.append(pinned)
            return
        time.sleep(0)
    W._PINNED_FREE.setdefault(2 * n_sums, []).append(pinned)


def _finish(s, sp, stream, bufs, n_sums, C, N, height, width, tile_height, tile_width, dev, needs_bwd, prefill):
    """From behind ``gs_step_fwd_begin`` to the compositing launch: the one host sync and both ``gs_step_fwd_finish`` phases."""
    i32, i64, f32, u8 = torch.int32, torch.int64, torch.float32, torch.uint8
    empty = torch.empty
    ptr = B.ptr
    n_elems = C * N
    pinned = bufs["pinned"]
    offsets = empty((C, tile_height, tile_width), dtype=i32, device=dev)
    s.offsets = ptr(offsets)
    sentinel = W._SentinelEvent(pinned, stream=torch.cuda.current_stream(dev))  # (the launch stream: what the wait watches for faults)
    # The phase-2 buffers are sized by n_isects, which only the read-back below delivers -- and everything the host does between the
    # read-back and the binning launch is time the GPU may spend idle (round 6, tools/host_timeline.py: three allocations, a size query
    # and the descriptor fields took 46 us there against the ~56 us of pre-sort the GPU still has when the sums land; 12-17 us of idle
    # GPU per step).  So they are allocated BEFORE the wait, with the capacity the same call shape needed last time (+ 3 %), and the
    # outputs are prefix views of them; a step that needs more falls back to exact allocations after the wait.
    key = (dev.index, C, N, tile_height, tile_width)
    cap = _ISECT_CAP.get(key, 0) if _PREALLOC else 0
    ids_buf = flat_buf = work = None
    wb = 0
    if cap:
        ids_buf = empty(cap, dtype=i64, device=dev)
        flat_buf = empty(cap, dtype=i32, device=dev)
        wb = B.query("gs_isect_finish_work_bytes", cap)
        work = empty(wb, dtype=u8, device=dev)
        s.isect_ids, s.flatten_ids, s.work, s.work_bytes = ptr(ids_buf), ptr(flat_buf), ptr(work), wb
    # ---- the one host sync: the count kernel's block sums land in pinned memory (-1 -> >= 0).  From here to the
    # binning launches the GPU has ~40 us of pre-sort left
    W._wait_event(sentinel)
    n_isects, n_kept_host = W.block_sum_totals(sentinel.np)
    s.n_kept_host = n_kept_host if W._PACKED_PAIRS else 0
    s.n_isects = n_isects
    if ids_buf is None or n_isects > cap:
        ids_buf = empty(n_isects, dtype=i64, device=dev)
        flat_buf = empty(n_isects, dtype=i32, device=dev)
        wb = B.query("gs_isect_finish_work_bytes", n_isects)
        work = empty(wb, dtype=u8, device=dev)
        s.isect_ids, s.flatten_ids, s.work, s.work_bytes = ptr(ids_buf), ptr(flat_buf), ptr(work), wb
    # the binning half goes out at once (the GPU has been waiting for this call since the pre-sort ended); the
    # compositing scratch is sized and allocated while it runs
    s.finish_phase = 1
    B.call("gs_step_fwd_finish", sp, stream)
    isect_ids = ids_buf if ids_buf.shape[0] == n_isects else ids_buf[:n_isects]
    flatten_ids = flat_buf if flat_buf.shape[0] == n_isects else flat_buf[:n_isects]
    _ISECT_CAP[key] = n_isects + (n_isects >> 5) + 4096
    W._PINNED_FREE.setdefault(2 * n_sums, []).append(pinned)
    bufs["pinned"] = None
    # ---- the compositing buffers (made while the GPU is busy with the binning)
    render_colors = empty((C, height, width, 3), dtype=f32, device=dev)
    render_alphas = empty((C, height, width, 1), dtype=f32, device=dev)
    last_ids = empty((C, height, width), dtype=i32, device=dev)
    fill = None
    if needs_bwd:
        extra = prefill.floats() if prefill is not None else 0
        if extra:
            extra += 64  # slack behind the last piece (a multi-GPU reduction rounds the span of all pieces up into it)
        fill = empty(n_elems * 16 + extra, dtype=f32, device=dev)
        if extra:
            prefill.carve(fill, n_elems * 16)
        else:
            prefill = None
    else:
        prefill = None
    s.render_colors, s.render_alphas, s.last_ids = ptr(render_colors), ptr(render_alphas), ptr(last_ids)
    if fill is not None:
        s.zero_fill, s.zero_fill_bytes = ptr(fill), fill.numel() * 4
    plan, sbytes = W._raster_plan(C * tile_height * tile_width, n_isects, 3, forward_only=not needs_bwd)
    ctypes.memmove(ctypes.addressof(s.plan), plan, 64)
    scratch = empty(sbytes, dtype=u8, device=dev)
    s.scratch = ptr(scratch)
    s.finish_phase = 2
    B.call("gs_step_fwd_finish", sp, stream)
    return offsets, isect_ids, flatten_ids, render_colors, render_alphas, last_ids, fill, prefill, scratch, plan


_ISECT_CAP: dict = {}  # (device, C, N, tile grid) -> capacity for the next call's intersection buffers (last count + 3 %)
_PREALLOC = os.environ.get("GS_ISECT_PREALLOC", "1") != "0"


class _StepProject(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, covars, quats, scales, viewmats, Ks, opacities, colors, sh_coeffs, sh_rest, mask_logits, backgrounds, cfg, hand,
                dyn_motion=None, dyn_omega=None, dyn_center=None, dyn_tscale=None, dyn=None):
        (width, height, eps2d, near_plane, far_plane, radius_clip, antialiased, camera_model, sh_degree, tile_size, tile_width,
         tile_height, needs_bwd, mask_cfg) = cfg
        mask_logits = _c(mask_logits)
        means, covars, quats, scales = _c(means), _c(covars), _c(quats), _c(scales)
        viewmats, Ks, opacities, colors = _c(viewmats), _c(Ks), _c(opacities), _c(colors)
        sh_coeffs, sh_rest, backgrounds = _c(sh_coeffs), _c(sh_rest), _c(backgrounds)
        C, N = viewmats.shape[0], means.shape[0]
        n_elems = C * N
        dev = means.device
        i32, i64, f32, u8 = torch.int32, torch.int64, torch.float32, torch.uint8
        empty = torch.empty
        s = _Step()
        s.C, s.N = C, N
        ptr = B.ptr
        s.means, s.covars, s.quats, s.scales, s.viewmats, s.Ks = ptr(means), ptr(covars), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks)
        s.opacities, s.colors, s.sh_coeffs, s.sh_rest, s.backgrounds = ptr(opacities), ptr(colors), ptr(sh_coeffs), ptr(sh_rest), ptr(backgrounds)
        s.sh_K = (sh_coeffs.shape[1] + (sh_rest.shape[1] if sh_rest is not None else 0)) if sh_coeffs is not None else 0
        s.sh_degree = int(sh_degree or 0)
        if mask_logits is not None:
            s.sh_mask_logits, s.sh_mask_temperature, s.sh_mask_binary = ptr(mask_logits), mask_cfg[0], int(mask_cfg[1])
        s.width, s.height, s.eps2d, s.near_plane, s.far_plane, s.radius_clip = width, height, eps2d, near_plane, far_plane, radius_clip
        cm = W._CAMERA_MODELS[camera_model]
        s.camera_model, s.antialiased = cm, int(antialiased)
        s.tile_size, s.tile_width, s.tile_height = tile_size, tile_width, tile_height
        ctx.dyn, ctx.dyn_first = None, 14
        if dyn is not None:  # dynamic splats: the slice (+ activations / round quantizer) inside the projection kernel
            dt = dyn.bind(quats, scales, opacities, colors, dyn_motion, dyn_omega, dyn_center, dyn_tscale)
            (s.dyn_motion, s.dyn_omega, s.dyn_trbf_center, s.dyn_trbf_scale, s.dyn_timestamp, s.dyn_raw_params, s.dyn_quant_mask,
             lo_, hi_, rng_, qn_) = dyn.c_args(dt)
            s.dyn_min_trbf, s.dyn_trbf_alive = dyn.min_trbf_arg(), ptr(dyn.alive_buffer(N, dev))
            for dst, src in ((s.dyn_quant_lo, dyn._tables[0]), (s.dyn_quant_hi, dyn._tables[1]), (s.dyn_quant_range, dyn._tables[2]),
                             (s.dyn_quant_step_norm, dyn._tables[3])):
                dst[:] = src[:]
            ctx.dyn = (dyn, dt)
        n_sums = C * ((N + 255) // 256)  # gs_projection_rows_blocks(N) per camera: the projection counts the tiles itself
        bufs = _phase1(s, C, N, dev, n_sums)
        radii, depths, rows, tiles_per_gauss = bufs["radii"], bufs["depths"], bufs["rows"], bufs["tiles_per_gauss"]
        stream = torch.cuda.current_stream(dev).cuda_stream
        sp = ctypes.addressof(s)
        prefill = None
        if needs_bwd and W.PREFILL_ENABLED:
            # the per-gaussian gradients the backward returns live behind the gradient rows in ONE zero-filled buffer
            # (_wrapper.GradPrefill: the compositing forward zero-fills it as a side job)
            need = ctx.needs_input_grad
            prefill = W.GradPrefill()
            prefill.request = W.prefill_request((
                ("means", means, need[0]), ("covars", covars, need[1]), ("quats", quats, need[2]), ("scales", scales, need[3]),
                ("opacities", opacities, need[6]), ("colors", colors, need[7]), ("sh", sh_coeffs, need[8]), ("sh_rest", sh_rest, need[9]))
                + (W.dyn_prefill_items(ctx.dyn, need, 14) if ctx.dyn is not None else ()))
        try:
            with torch.cuda.device(dev):
                B.call("gs_step_fwd_begin", sp, stream)
                (offsets, isect_ids, flatten_ids, render_colors, render_alphas, last_ids, fill, prefill, scratch, plan) = _finish(
                    s, sp, stream, bufs, n_sums, C, N, height, width, tile_height, tile_width, dev, needs_bwd, prefill)
        except BaseException:
            _drop_pinned(bufs, n_sums)  # (an error between the two calls: the count kernel may still be storing into it)
            raise
        # ---- node 2's share
        hand.render_colors, hand.render_alphas, hand.last_ids, hand.scratch, hand.plan = render_colors, render_alphas, last_ids, scratch, plan
        hand.grad_rows = fill[:n_elems * 16].view(C, N, 16) if fill is not None else None
        hand.offsets, hand.flatten_ids = offsets, flatten_ids
        # ---- this node's backward is _ProjectRows.backward: same saved tensors, same attributes
        ctx.save_for_backward(means, covars, quats, scales, viewmats, Ks, opacities, radii, rows, sh_coeffs, sh_rest)
        ctx.width, ctx.height, ctx.eps2d, ctx.cm, ctx.antialiased = width, height, eps2d, cm, bool(antialiased)
        ctx.has_colors, ctx.sh_degree = colors is not None, (int(sh_degree) if sh_coeffs is not None else None)
        ctx.prefill = prefill
        ctx.mask = (mask_logits, float(mask_cfg[0]), bool(mask_cfg[1])) if mask_logits is not None else None
        ctx.mark_non_differentiable(radii, rows, tiles_per_gauss, isect_ids, flatten_ids, offsets)
        ctx.set_materialize_grads(False)
        R = W
        return (radii, rows[..., R.ROW_MEAN2D:R.ROW_MEAN2D + 2], depths, rows[..., R.ROW_CONIC:R.ROW_CONIC + 3], rows[..., R.ROW_OPACITY],
                rows[..., R.ROW_COLOR:R.ROW_COLOR + 3], rows, tiles_per_gauss, isect_ids, flatten_ids, offsets)

    @staticmethod
    def backward(ctx, v_radii, v_means2d, v_depths, v_conics, v_opac, v_colors, v_rows, *_ints):
        g = W._ProjectRows.backward(ctx, v_radii, v_means2d, v_depths, v_conics, v_opac, v_colors, v_rows)
        # _ProjectRows' inputs: means, covars, quats, scales, viewmats, Ks, opacities, colors, sh_coeffs, sh_rest, ...
        head = (g[0], g[1], g[2], g[3], g[4], None, g[6], g[7], g[8], g[9], g[10], None, None, None)
        return head + tuple(g[22:27]) if ctx.dyn is not None else head


class _StepComposite(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, cfg, hand):
        width, height, tile_size, absgrad = cfg
        ctx.grad_rows = hand.grad_rows
        ctx.plan, ctx.strides = hand.plan, _ROW_STRIDES
        ctx.save_for_backward(means2d, conics, colors, opacities, backgrounds, None, hand.offsets, hand.flatten_ids, hand.render_alphas,
                              hand.last_ids, hand.scratch, hand.render_colors)
        ctx.width, ctx.height, ctx.tile_size, ctx.absgrad, ctx.deterministic = width, height, tile_size, absgrad, False
        ctx.set_materialize_grads(False)
        return hand.render_colors, hand.render_alphas

    @staticmethod
    def backward(ctx, v_render_colors, v_render_alphas):
        g = W._RasterizeToPixels.backward(ctx, v_render_colors, v_render_alphas)
        return (g[0], g[1], g[2], g[3], g[4], None, None)


class _RowsState:
    """Binning in flight over splat rows some other producer wrote (``rows_begin`` ... ``rows_composite`` | ``rows_abandon``)."""

    __slots__ = ("s", "bufs", "n_sums", "C", "N", "dev", "tile")

    def __del__(self):  # dropped without finish or abandon (an exception in between): same care for the pinned buffer
        try:
            _drop_pinned(getattr(self, "bufs", None), getattr(self, "n_sums", 0))
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass


def rows_applicable(rows: Optional[Tensor], colors: Optional[Tensor], packed: bool, render_mode: str, channel_chunk: int,
                    deterministic: bool, absgrad: bool) -> bool:
    """The gaussian-sharded mode's receiver side: [C_local, N_total, 16] splat rows with RGB colours in them."""
    return (ENABLED and rows is not None and rows.is_cuda and not packed and not deterministic and render_mode == "RGB"
            and channel_chunk >= 3 and colors is not None and colors.shape[-1] == 3
            # (the block sums go straight into pinned memory: same size gate as the one-GPU fast path / isect_tiles_begin --
            # tens of thousands of direct PCIe stores stalled the GPU for ~85 ms on some forwards)
            and 0 < rows.shape[0] * rows.shape[1] <= W._PINNED_DIRECT_MAX * 1024)


def rows_begin(radii: Tensor, depths: Tensor, rows: Tensor, tile_size: int, tile_width: int, tile_height: int) -> _RowsState:
    """Queue the binning of ``rows`` up to its host read-back (``gs_step_fwd_begin`` with ``rows_ready``): count + depth keys
    -> depth pre-sort.  Nothing differentiable happens here (the reference's ``isect_tiles`` is not differentiable either)."""
    C, N = radii.shape
    dev = rows.device
    assert rows.is_contiguous() and radii.is_contiguous() and depths.is_contiguous() and rows.shape == (C, N, W.ROW)
    st = _RowsState()
    s = _Step()
    s.C, s.N, s.rows_ready = C, N, 1
    s.tile_size, s.tile_width, s.tile_height = tile_size, tile_width, tile_height
    n_sums = int(B.query("gs_isect_count_blocks", C * N))
    st.s, st.n_sums, st.C, st.N, st.dev, st.tile = s, n_sums, C, N, dev, (tile_size, tile_width, tile_height)
    st.bufs = None
    st.bufs = _phase1(s, C, N, dev, n_sums, given=(radii, depths, rows))
    with torch.cuda.device(dev):
        B.call("gs_step_fwd_begin", ctypes.addressof(s), torch.cuda.current_stream(dev).cuda_stream)
    return st


def rows_abandon(st: Optional[_RowsState]) -> None:
    """Drop a ``rows_begin`` without finishing it (the sparse exchange's overflow retry).  The count kernel stores its block
    sums straight into the pinned buffer: it goes back to the free list only once the kernel has run."""
    if st is not None:
        _drop_pinned(st.bufs, st.n_sums)


class _StepRowsComposite(torch.autograd.Function):
    """Second half over rows: host sync -> emit + pair sort + offsets -> compositing forward; backward =
    ``_RasterizeToPixels.backward`` (the gradient rows feed the exchange's backward)."""

    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, cfg, st, prefill):
        width, height, absgrad = cfg
        s, C, N, dev = st.s, st.C, st.N, st.dev
        tile_size, tile_width, tile_height = st.tile
        backgrounds = _c(backgrounds)
        s.backgrounds = B.ptr(backgrounds)
        s.width, s.height = width, height
        needs_bwd = any(ctx.needs_input_grad[:5])
        stream = torch.cuda.current_stream(dev).cuda_stream
        with torch.cuda.device(dev):
            (offsets, isect_ids, flatten_ids, render_colors, render_alphas, last_ids, fill, prefill, scratch, plan) = _finish(
                s, ctypes.addressof(s), stream, st.bufs, st.n_sums, C, N, height, width, tile_height, tile_width, dev, needs_bwd, prefill)
        ctx.grad_rows = fill[:C * N * 16].view(C, N, 16) if fill is not None else None
        ctx.plan, ctx.strides = plan, _ROW_STRIDES
        ctx.save_for_backward(means2d, conics, colors, opacities, backgrounds, None, offsets, flatten_ids, render_alphas, last_ids, scratch,
                              render_colors)
        ctx.width, ctx.height, ctx.tile_size, ctx.absgrad, ctx.deterministic = width, height, tile_size, absgrad, False
        tiles_per_gauss = st.bufs["tiles_per_gauss"]
        ctx.mark_non_differentiable(tiles_per_gauss, isect_ids, flatten_ids, offsets)
        ctx.set_materialize_grads(False)
        st.bufs = None  # (its pinned buffer went back to the free list inside _finish)
        return render_colors, render_alphas, tiles_per_gauss, isect_ids, flatten_ids, offsets

    @staticmethod
    def backward(ctx, v_render_colors, v_render_alphas, *_ints):
        g = W._RasterizeToPixels.backward(ctx, v_render_colors, v_render_alphas)
        return (g[0], g[1], g[2], g[3], g[4], None, None, None)


def rows_composite(st: _RowsState, means2d, conics, colors, opacities, backgrounds, width, height, absgrad, prefill):
    """-> (render_colors, render_alphas, tiles_per_gauss, isect_ids, flatten_ids, isect_offsets)"""
    return _StepRowsComposite.apply(means2d, conics, colors, opacities, backgrounds, (int(width), int(height), bool(absgrad)), st, prefill)


_GSHIFT, _PREFIX_FROM = [0], [8192]


def _init_consts() -> int:
    check_layout()  # (once per process, before the first descriptor is handed to the library)
    _GSHIFT[0] = int(B.query("gs_isect_emit_group_shift"))
    _PREFIX_FROM[0] = int(B.query("gs_isect_emit_prefix_from_groups"))
    return _GSHIFT[0]


def rasterize_step(means, covars, quats, scales, opacities, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip,
                   antialiased, camera_model, row_colors, sh_coeffs, sh_rest, sh_degree, tile_size, backgrounds, absgrad, sh_mask=None,
                   dynamic=None):
    """The fast path's forward; returns ``(render_colors, render_alphas, meta)`` with the reference's meta keys."""
    C, N = viewmats.shape[0], means.shape[0]
    tile_width, tile_height = math.ceil(width / float(tile_size)), math.ceil(height / float(tile_size))
    mask_logits = sh_mask[0] if sh_mask is not None else None
    dyn_in = ()
    if dynamic is not None:
        dynamic.check(N)
        dyn_in = (dynamic.motion, dynamic.omega, dynamic.trbf_center, dynamic.trbf_scale)
    needs_bwd = torch.is_grad_enabled() and any(
        t is not None and t.requires_grad for t in (means, covars, quats, scales, opacities, row_colors, sh_coeffs, sh_rest, backgrounds,
                                                    mask_logits) + dyn_in)
    hand = _Handover()
    cfg = (int(width), int(height), float(eps2d), float(near_plane), float(far_plane), float(radius_clip), bool(antialiased), camera_model,
           sh_degree, int(tile_size), tile_width, tile_height, needs_bwd,
           (float(sh_mask[1]), bool(sh_mask[2])) if sh_mask is not None else None)
    (radii, means2d, depths, conics, opac_cn, colors_cn, rows, tiles_per_gauss, isect_ids, flatten_ids, offsets) = _StepProject.apply(
        means, covars, quats, scales, viewmats, Ks, opacities, row_colors, sh_coeffs, sh_rest, mask_logits, backgrounds, cfg, hand,
        *(dyn_in + (dynamic,) if dynamic is not None else ()))
    render_colors, render_alphas = _StepComposite.apply(means2d, conics, colors_cn, opac_cn, backgrounds,
                                                        (int(width), int(height), int(tile_size), bool(absgrad)), hand)
    meta = {
        "camera_ids": None, "gaussian_ids": None, "radii": radii, "means2d": means2d, "depths": depths, "conics": conics,
        "opacities": opac_cn, "tile_width": tile_width, "tile_height": tile_height, "tiles_per_gauss": tiles_per_gauss,
        "isect_ids": isect_ids, "flatten_ids": flatten_ids, "isect_offsets": offsets, "width": width, "height": height,
        "tile_size": tile_size, "n_cameras": C,
    }
    return render_colors, render_alphas, meta
