"""Render API: ``rasterization()`` with the reference's signature, defaults, return values
and ``meta`` dictionary (reference ``gsplat/rendering.py:28-582``), orchestrating the five
HIP stages  project -> SH colour -> tile-intersect + radix sort -> offset-encode ->
per-tile alpha compositing.

Drop-in contract kept from the reference:
* ``means2d`` stays an autograd intermediate between the projection Function and the
  rasterize Function, so ``meta["means2d"].retain_grad()`` / ``.absgrad`` work for the
  densification strategies (reference strategy/default.py:150, 221-226);
* ``meta`` keys and dtypes: camera_ids, gaussian_ids, radii, means2d, depths, conics,
  opacities, tile_width, tile_height, tiles_per_gauss, isect_ids, flatten_ids,
  isect_offsets, width, height, tile_size, n_cameras.

Differences by design: shared SH coefficients are not expanded to ``[C,N,K,3]`` and
channel padding is not needed (see _wrapper.py).
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor
from typing_extensions import Literal

from . import _step
from .compression_simulation.ada_mask import MaskedShN
from ._wrapper import (
    ROW_COLOR,
    fully_fused_projection,
    project_rows,
    GradPrefill,
    gather_rows,
    isect_offset_encode,
    isect_tiles,
    isect_tiles_abandon,
    isect_tiles_begin,
    isect_tiles_finish,
    isect_tiles_start,
    rasterize_to_pixels,
    spherical_harmonics,
    spherical_harmonics_shared,
    spherical_harmonics_view,
)


def _camera_centers(viewmats: Tensor) -> Tensor:
    """Camera positions in world space, ``inverse(viewmats)[:, :3, 3]``, in closed form.

    The reference calls ``torch.inverse(viewmats)`` (rendering.py:370); on ROCm that goes through
    a batched LU with a host-side status check, i.e. a device synchronisation of ~0.5 ms in the
    middle of every step (measured: tools/cpu_prof.py).  For an affine world->camera matrix
    [[A, t], [0, 1]] the inverse's translation is -A^-1 t, and A^-1 = adj(A) / det(A) is three
    cross products -- a handful of tiny asynchronous kernels, differentiable through autograd.
    """
    if viewmats.is_cuda and not viewmats.requires_grad:
        from . import _backend as B

        vm = viewmats.contiguous()
        out = torch.empty((vm.shape[0], 3), dtype=torch.float32, device=vm.device)
        with torch.cuda.device(vm.device):
            B.call("gs_camera_centers", vm.shape[0], B.ptr(vm), B.ptr(out), torch.cuda.current_stream(vm.device).cuda_stream)
        return out
    A = viewmats[:, :3, :3]
    t = viewmats[:, :3, 3]
    c0, c1, c2 = A[:, :, 0], A[:, :, 1], A[:, :, 2]
    r0 = torch.linalg.cross(c1, c2)
    r1 = torch.linalg.cross(c2, c0)
    r2 = torch.linalg.cross(c0, c1)
    det = (c0 * r0).sum(-1, keepdim=True)
    inv_t = torch.stack([(r0 * t).sum(-1), (r1 * t).sum(-1), (r2 * t).sum(-1)], dim=-1) / det
    return -inv_t


def _prefill_enabled() -> bool:
    from . import _wrapper

    return _wrapper.PREFILL_ENABLED


def _step_max_elems() -> int:
    from ._wrapper import _PINNED_DIRECT_MAX

    return _PINNED_DIRECT_MAX * 1024  # (the count kernel's block sums go straight into pinned memory up to this size)


class _RowsColorDepth(torch.autograd.Function):
    """``cat((colors, depths[..., None]), -1)`` when ``colors`` are columns 6:9 of the splat rows: column 9 of the same rows holds the
    depth (include/gsplat_hip.h "Splat rows"), so the four channels are the view ``rows[..., 6:10]`` -- nothing is copied, and on the
    way back the compositing backward's gradient rows stay the one buffer the projection backward reads in place."""

    @staticmethod
    def forward(ctx, colors, depths, hold):
        ctx.set_materialize_grads(False)
        return hold[0][..., ROW_COLOR:ROW_COLOR + 4]

    @staticmethod
    def backward(ctx, v):
        if v is None:
            return None, None, None
        return v[..., :3], v[..., 3], None


class _ExpectedDepth(torch.autograd.Function):
    """The tail of the "ED" / "RGB+ED" modes in one kernel each way (``gs_expected_depth_fwd`` / ``_bwd``)."""

    @staticmethod
    def forward(ctx, renders, alphas):
        from . import _backend as B

        renders, alphas = renders.contiguous(), alphas.contiguous()
        out = torch.empty_like(renders)
        n_pix, ch = alphas.numel(), renders.shape[-1]
        with torch.cuda.device(renders.device):
            B.call("gs_expected_depth_fwd", n_pix, ch, B.ptr(renders), B.ptr(alphas), B.ptr(out), torch.cuda.current_stream(renders.device).cuda_stream)
        ctx.save_for_backward(renders, alphas)
        return out

    @staticmethod
    def backward(ctx, v_out):
        from . import _backend as B

        renders, alphas = ctx.saved_tensors
        v_out = v_out.contiguous().float()
        need = ctx.needs_input_grad
        v_r = torch.empty_like(renders) if need[0] else None
        v_a = torch.empty_like(alphas) if need[1] else None
        with torch.cuda.device(renders.device):
            B.call("gs_expected_depth_bwd", alphas.numel(), renders.shape[-1], B.ptr(renders), B.ptr(alphas), B.ptr(v_out), B.ptr(v_r), B.ptr(v_a),
                   torch.cuda.current_stream(renders.device).cuda_stream)
        return v_r, v_a


def rasterization(
    means: Tensor,  # [N, 3]
    quats: Tensor,  # [N, 4]
    scales: Tensor,  # [N, 3]
    opacities: Tensor,  # [N]
    colors,  # Tensor [(C,) N, D] or [(C,) N, K, 3]; or the pair (sh0 [N, 1, 3], shN [N, K-1, 3]) with sh_degree
    viewmats: Tensor,  # [C, 4, 4]
    Ks: Tensor,  # [C, 3, 3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = True,
    tile_size: int = 16,
    backgrounds: Optional[Tensor] = None,
    render_mode: Literal["RGB", "D", "ED", "RGB+D", "RGB+ED"] = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: Literal["classic", "antialiased"] = "classic",
    channel_chunk: int = 32,
    distributed: bool = False,
    camera_model: Literal["pinhole", "ortho", "fisheye"] = "pinhole",
    covars: Optional[Tensor] = None,
    deterministic: bool = False,
    dynamic=None,
) -> Tuple[Tensor, Tensor, Dict]:
    """Rasterize a set of 3D Gaussians (N) to a batch of image planes (C).

    Same semantics as the reference ``gsplat.rendering.rasterization``.

    Args:
        means: 3D centres [N,3].  quats: rotations wxyz [N,4] (need not be normalised).
        scales: [N,3].  opacities: [N].  colors: [(C,)N,D] post-activation features, or SH
        coefficients [(C,)N,K,3] when ``sh_degree`` is set.  viewmats: world->camera
        [C,4,4].  Ks: intrinsics [C,3,3].  covars: optional [N,3,3] replacing quats/scales.
        packed: COO intermediates (memory-efficient for many cameras).  sparse_grad: COO
        gradients (needs packed).  absgrad: also accumulate |d means2d| into
        ``meta["means2d"].absgrad``.  rasterize_mode "antialiased": opacity compensation.
        render_mode: RGB / D / ED / RGB+D / RGB+ED.  distributed: gaussian-sharded
        multi-GPU mode of the reference (see distributed.py; the camera-sharded data
        parallel wrapper is ``distributed.rasterization_camera_sharded``).
        deterministic (opt-in, beyond the reference's signature): bit-reproducible gradients -- the compositing
        backward accumulates in fixed point instead of with float atomics (up to 4 render channels).
        dynamic (opt-in, beyond the reference's signature): a ``dynamic.DynamicSlice`` or the tuple ``(motion [N,9], omega [N,4],
        trbf_center [N,1], trbf_scale [N,1], timestamp)`` -- the splats are DYNAMIC (spacetime) gaussians and ``means`` / ``quats`` /
        ``opacities`` their time-independent parameters: the temporal slice of the reference's dynamic trainer
        (examples/simple_trainer_dyngs.py:506-521) is evaluated inside the projection kernels, bit-identical to
        ``dynamic.temporal_slice`` followed by this call (csrc/projection_dyn.hip; routes the fused kernels do not cover take
        exactly that detour).

    Returns:
        render_colors [C,H,W,X], render_alphas [C,H,W,1], meta dict.
    """
    meta: Dict = {}

    N = means.shape[0]
    C = viewmats.shape[0]
    assert C > 0, "rasterization needs at least one camera (the reference fails on an empty batch too: log2(0) in isect_tiles)"
    device = means.device
    if dynamic is not None:
        from .dynamic import DynamicSlice

        dynamic = DynamicSlice.of(dynamic)
        dynamic.check(N)
        assert covars is None, "dynamic splats rotate their quaternions in time: pass quats + scales, not covars"
        dyn_fused = (not packed and not distributed and covars is None and sh_degree is None and means.is_cuda and viewmats.is_cuda
                     and not viewmats.requires_grad and torch.is_tensor(colors) and colors.dim() == 2)
        if "colors" in dynamic.quantize and not (dyn_fused and colors.shape[-1] == 3):
            dyn_fused = False
        if not dyn_fused:  # every other route: the same chain through the stand-alone operators, then the plain call
            means, quats, scales, opacities, c_ = dynamic.apply_unfused(means, quats, scales, opacities,
                                                                         colors if torch.is_tensor(colors) else None)
            colors = c_ if c_ is not None else colors
            dynamic = None
    assert means.shape == (N, 3), means.shape
    if covars is None:
        assert quats.shape == (N, 4), quats.shape
        assert scales.shape == (N, 3), scales.shape
    else:
        assert covars.shape == (N, 3, 3), covars.shape
        quats, scales = None, None
        # 3x3 matrix -> upper-triangular 6-vector
        covars = covars[..., [0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2]]
    assert opacities.shape == (N,), opacities.shape
    assert viewmats.shape == (C, 4, 4), viewmats.shape
    assert Ks.shape == (C, 3, 3), Ks.shape
    assert render_mode in ["RGB", "D", "ED", "RGB+D", "RGB+ED"], render_mode
    # Opt-in beyond the reference's signature: SH coefficients as the PAIR (sh0, shN) the trainers keep as separate parameters
    # (reference examples/simple_trainer.py:779-786 concatenates them before every render: 193 MB each way at 1 M splats, and
    # autograd splits the gradient again).  The fused route takes the two tensors as they are; every other route gets the cat.
    sh_rest = None
    sh_mask = None
    if isinstance(colors, (tuple, list)):
        assert sh_degree is not None and len(colors) == 2, "a (sh0, shN) pair needs sh_degree"
        sh0, shN = colors
        masked = shN if isinstance(shN, MaskedShN) else None  # shN + the mask to apply to it (compression_simulation.ada_mask)
        if masked is not None:
            shN = masked.shN
        assert sh0.shape == (N, 1, 3) and shN.dim() == 3 and shN.shape[0] == N and shN.shape[2] == 3, (sh0.shape, shN.shape)
        split_ok = ((not packed) and (not distributed) and means.is_cuda and viewmats.is_cuda and not viewmats.requires_grad
                    and shN.shape[1] >= 1)
        # the fused mask rides on the fused SH backward: vectorisable rows (3 K % 4 == 0), which covers degrees 1 and 3
        # -- and everything else the backward's fused route checks must be known to hold HERE, or the forward would succeed and
        # training die in loss.backward() (GS_FUSE_SH_BWD=0, a misaligned shN view): such masks are materialised up front
        from ._wrapper import _FUSE_SH_BWD
        mask_ok = masked is None or (split_ok and _FUSE_SH_BWD and (3 * (1 + shN.shape[1])) % 4 == 0 and shN.is_contiguous()
                                     and shN.data_ptr() % 16 == 0 and masked.mask_logits.numel() == N)
        if masked is not None and not mask_ok:
            shN, masked = masked.materialize(), None
        if split_ok:
            colors, sh_rest = sh0, shN
            if masked is not None:
                sh_mask = (masked.mask_logits, masked.temperature, masked.binary)
        else:
            colors = torch.cat([sh0, shN], dim=1)
    # the compositing kernels map a tile onto wave64 quadrants of 8x8 pixels: tiles up to 16x16 (the reference launches
    # tile_size^2 threads per block, i.e. accepts up to 32; every caller in the reference uses 16).  Checked here, before
    # projection and binning run, instead of surfacing as a native error afterwards.
    assert 1 <= tile_size <= 32 and (tile_size <= 16 or tile_size % 2 == 0), \
        f"tile_size must be in [1, 16] or an even size up to 32 on the HIP backend, got {tile_size}"

    if sh_degree is None:
        # post-activation values [N, D] or [C, N, D]
        assert (colors.dim() == 2 and colors.shape[0] == N) or (
            colors.dim() == 3 and colors.shape[:2] == (C, N)
        ), colors.shape
        if distributed:
            assert colors.dim() == 2, "Distributed mode only supports per-Gaussian colors."
    else:
        # SH coefficients [N, K, 3] or [C, N, K, 3]; partial bands allowed
        assert (colors.dim() == 3 and colors.shape[0] == N and colors.shape[2] == 3) or (
            colors.dim() == 4 and colors.shape[:2] == (C, N) and colors.shape[3] == 3
        ), colors.shape
        assert (sh_degree + 1) ** 2 <= colors.shape[-2] + (sh_rest.shape[1] if sh_rest is not None else 0), colors.shape
        if distributed:
            assert colors.dim() == 3, "Distributed mode only supports per-Gaussian colors."

    if absgrad:
        assert not distributed, "AbsGrad is not supported in distributed mode."

    if distributed:
        from . import distributed as D

        world_rank = torch.distributed.get_rank()
        world_size = torch.distributed.get_world_size()
        # gaussians are sharded over ranks; gather #gaussians and all cameras
        C_world = [C] * world_size
        cap_world = None  # chunk capacities of the sparse exchange (None: every row travels)
        if viewmats.requires_grad or Ks.requires_grad:
            N_world = D.all_gather_int32(world_size, N, device=device)
            viewmats, Ks = D.all_gather_tensor_list(world_size, [viewmats, Ks])
        else:
            sparse = (not packed) and D.sparse_enabled(means, C_world)
            # (the shard sizes travel over the host group while the projection is queued: resolved at the exchange)
            shard_sizes, viewmats, Ks = D.gather_shard_meta(world_size, N, viewmats, Ks, D.sparse_capacity(C, N) if sparse else 0)
            N_world = None
            cap_world = () if sparse else None  # chunk capacities: known to be in use, values still in flight
        C = len(viewmats)

    # Unpacked batches on one GPU go through SPLAT ROWS: the projection writes one 64-byte row per (camera, gaussian) --
    # mean2d, conic, opacity (x antialias compensation), colour, depth, radius -- that the compositing kernels fetch whole;
    # means2d / conics / opacities (/ colours) below are column views of that buffer (same shapes and dtypes as the
    # reference's separate tensors; like its means2d / conics they are only defined where radii > 0).
    fuse_sh = (sh_degree is not None and not packed and colors.dim() == 3 and not viewmats.requires_grad and viewmats.is_cuda)
    # (gaussian-sharded: when the colours sit in the rows and the sparse exchange is on, the rows themselves travel --
    # distributed._ExchangeRows; otherwise the separate arrays of the reference's layout do)
    dist_rows = (distributed and cap_world is not None and means.is_cuda
                 and (fuse_sh or (sh_degree is None and colors.dim() == 2 and colors.shape[-1] == 3)))
    use_rows = (not packed) and means.is_cuda and (not distributed or dist_rows)
    # the fused SH route reads the means a second time (view directions): it gets them back FROM the projection, so that
    # its contribution to d/d means is added inside the projection's backward kernel
    means_alias = fuse_sh and means.requires_grad and not use_rows
    rows = None
    compensations = None
    prefill = None
    if use_rows:
        row_colors = colors if (sh_degree is None and colors.dim() == 2 and colors.shape[-1] == 3) else None
        if _step.applicable(means, viewmats, colors, sh_degree, packed, distributed, render_mode, channel_chunk, deterministic,
                            fuse_sh, row_colors) and C * N <= _step_max_elems() and tile_size <= 16:
            # the common training call: the whole forward as two native calls around the one host read-back (_step.py)
            return _step.rasterize_step(
                means, covars, quats, scales, opacities, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip,
                rasterize_mode == "antialiased", camera_model, row_colors, colors if fuse_sh else None, sh_rest,
                sh_degree if fuse_sh else None, tile_size, backgrounds, absgrad, sh_mask=sh_mask, dynamic=dynamic)
        # the dense per-gaussian gradients of the projection node are allocated and zero-filled by the compositing
        # forward's side job; its backward then writes the visible gaussians' rows only (_wrapper.GradPrefill)
        prefill = GradPrefill() if (torch.is_grad_enabled() and _prefill_enabled()) else None
        radii, means2d, depths, conics, opacities, colors_rows, rows = project_rows(
            means, covars, quats, scales, viewmats, Ks, width, height, opacities, row_colors,
            eps2d=eps2d, near_plane=near_plane, far_plane=far_plane, radius_clip=radius_clip,
            antialiased=(rasterize_mode == "antialiased"), camera_model=camera_model,
            # shared SH coefficients and fixed poses: the colours are evaluated by the projection pass itself
            sh_coeffs=colors if fuse_sh else None, sh_degree=sh_degree if fuse_sh else None, sh_rest=sh_rest,
            prefill=prefill, sh_mask=sh_mask, dynamic=dynamic,
        )
        camera_ids, gaussian_ids = None, None
        opacity_rider = False
        if row_colors is not None or fuse_sh:
            colors = colors_rows  # [C, N, 3]: columns 6:9 of the rows
    else:
        proj_results = fully_fused_projection(
            means, covars, quats, scales, viewmats, Ks, width, height,
            eps2d=eps2d, packed=packed, near_plane=near_plane, far_plane=far_plane,
            radius_clip=radius_clip, sparse_grad=sparse_grad,
            calc_compensations=(rasterize_mode == "antialiased"), camera_model=camera_model, _means_alias=means_alias,
        )
        means_sh = means
        if means_alias:
            means_sh, proj_results = proj_results[5], proj_results[:5]

        if packed:
            camera_ids, gaussian_ids, radii, means2d, depths, conics, compensations = proj_results
            opacities = gather_rows(opacities, gaussian_ids)  # [nnz] = opacities[gaussian_ids], one-pass atomic backward
        else:
            radii, means2d, depths, conics, compensations = proj_results
            camera_ids, gaussian_ids = None, None
            # classic mode + shared SH on the fused route: the per-view opacities ride along with the colour kernels
            # (written by the SH forward, summed over cameras by its backward) instead of `.repeat` + autograd's sum
            opacity_rider = (compensations is None and sh_degree is not None and colors.dim() == 3
                             and not viewmats.requires_grad and viewmats.is_cuda)
            opacities_n = opacities
            if not opacity_rider:
                opacities = opacities.repeat(C, 1)  # [C, N]

        if compensations is not None:
            opacities = opacities * compensations

    meta.update(
        {
            "camera_ids": camera_ids,
            "gaussian_ids": gaussian_ids,
            "radii": radii,
            "means2d": means2d,
            "depths": depths,
            "conics": conics,
            "opacities": opacities,
        }
    )

    tile_width = math.ceil(width / float(tile_size))
    tile_height = math.ceil(height / float(tile_size))
    # Tile binning is split around its one host sync (the intersection count): the first half is queued here, the
    # colour evaluation below then runs on the GPU while the host waits for the count.
    isect_state = None
    if not distributed:
        n_elems = int(radii.numel())
        isect_state = isect_tiles_begin(means2d, radii, depths, tile_size, tile_width, tile_height, True, C,
                                        0 if packed else N, n_elems, camera_ids.contiguous() if packed else None)

    # colours -> [C, N, D] or [nnz, D]
    if use_rows and fuse_sh:
        pass  # evaluated by the projection pass (columns 6:9 of the splat rows)
    elif sh_degree is None:
        if packed:
            colors = gather_rows(colors, gaussian_ids) if colors.dim() == 2 else colors[camera_ids, gaussian_ids]
        else:
            if colors.dim() == 2:
                # (one camera: a view -- autograd's expand backward is a sum over the camera axis, a 35 us reduce kernel at 2 M x 9
                # floats even when that axis has one entry)
                colors = colors[None] if C == 1 else colors.expand(C, -1, -1)
    else:
        fused_sh = False
        fuse = fuse_sh
        campos = None if fuse else _camera_centers(viewmats)  # [C, 3] == inverse(viewmats)[:, :3, 3]
        if packed:
            dirs = gather_rows(means, gaussian_ids) - campos[camera_ids]  # [nnz, 3]
            masks = radii > 0
            shs = gather_rows(colors, gaussian_ids) if colors.dim() == 3 else colors[camera_ids, gaussian_ids, :, :]
            colors = spherical_harmonics(sh_degree, dirs, shs, masks=masks)  # [nnz, 3]
        else:
            if fuse:
                # fused: camera centres, dirs, mask, SH and clamp_min(. + 0.5, 0) in one kernel each way
                if opacity_rider:
                    colors, opacities = spherical_harmonics_view(sh_degree, means_sh, viewmats, colors, radii, opacities=opacities_n)
                    meta["opacities"] = opacities
                else:  # (with splat rows: written into columns 6:9 of the rows, returned as that view)
                    colors = spherical_harmonics_view(sh_degree, means_sh, viewmats, colors, radii, rows=rows)  # [C, N, 3]
                fused_sh = True
            else:
                dirs = means[None, :, :] - campos[:, None, :]  # [C, N, 3]
                masks = radii > 0  # [C, N]
                if colors.dim() == 3:
                    colors = spherical_harmonics_shared(sh_degree, dirs, colors, masks=masks)  # [C, N, 3]
                else:
                    colors = spherical_harmonics(sh_degree, dirs, colors, masks=masks)  # [C, N, 3]
        # same convention as the reference (rendering.py:392)
        if not fused_sh:
            colors = torch.clamp_min(colors + 0.5, 0.0)

    if distributed:
        from . import distributed as D

        # redistribute, then bin.  The sparse exchange sizes its chunks from earlier steps without a read-back; should a
        # chunk have been too small (every rank learns it at the tile-count read-back), repeat once at full capacity.
        pre = (radii, means2d, depths, conics, opacities, colors, camera_ids, gaussian_ids)
        pre_rows = rows
        if N_world is None:
            N_world, caps = shard_sizes()
            cap_world = caps if cap_world is not None else None
        for attempt in range(2):
            if use_rows:
                C, radii, means2d, depths, conics, opacities, colors, rows = D.exchange_rows(
                    world_rank, N, N_world, C_world, cap_world, *pre[:6], pre_rows)
            else:
                (C, radii, means2d, depths, conics, opacities, colors, camera_ids, gaussian_ids) = D.exchange_projected(
                    world_rank, world_size, N, N_world, C_world, packed, *pre, cap_world=cap_world,
                )
            # binning up to its read-back; the depth pre-sort queued behind the count keeps the GPU busy while the host
            # looks at the overflow flags (stored to pinned memory before the count) and comes back for the rest
            rows_state = None
            if use_rows and tile_size <= 16 and _step.rows_applicable(rows, colors, packed, render_mode, channel_chunk, deterministic, absgrad):
                # (received rows: binning + compositing as native calls around the read-back, like the one-GPU fast path)
                rows_state = _step.rows_begin(radii, depths, rows, tile_size, tile_width, tile_height)
            else:
                isect_state = isect_tiles_start(
                    means2d, radii, depths, tile_size, tile_width, tile_height,
                    packed=packed, n_cameras=C, camera_ids=camera_ids, gaussian_ids=gaussian_ids,
                )
            if cap_world is None or not D.exchange_overflowed():
                break
            # (the pinned block-sum buffer is still being written by the count kernel)
            _step.rows_abandon(rows_state)
            isect_tiles_abandon(isect_state)
            isect_state = rows_state = None
            cap_world = [C_world[r] * N_world[r] for r in range(world_size)]
        if rows_state is not None:
            render_colors, render_alphas, tiles_per_gauss, isect_ids, flatten_ids, isect_offsets = _step.rows_composite(
                rows_state, means2d, conics, colors, opacities, backgrounds, width, height, absgrad, prefill)
            meta.update({"tile_width": tile_width, "tile_height": tile_height, "tiles_per_gauss": tiles_per_gauss, "isect_ids": isect_ids,
                         "flatten_ids": flatten_ids, "isect_offsets": isect_offsets, "width": width, "height": height,
                         "tile_size": tile_size, "n_cameras": C})
            return render_colors, render_alphas, meta

    if render_mode in ["RGB+D", "RGB+ED"]:
        if (rows is not None and colors.dim() == 3 and colors.shape[-1] == 3 and colors.data_ptr() == rows.data_ptr() + 4 * ROW_COLOR
                and colors.stride() == rows.stride()[:-1] + (1,) and not distributed):
            # the colours ride in the splat rows, whose next column IS the depth: the four channels are a view (no cat, and the
            # compositing backward's gradient rows reach the projection backward in place instead of through autograd's split)
            colors = _RowsColorDepth.apply(colors, depths, (rows,))
        else:
            colors = torch.cat((colors, depths[..., None]), dim=-1)
        if backgrounds is not None:
            backgrounds = torch.cat([backgrounds, torch.zeros(C, 1, device=backgrounds.device)], dim=-1)
    elif render_mode in ["D", "ED"]:
        colors = depths[..., None]
        if backgrounds is not None:
            backgrounds = torch.zeros(C, 1, device=backgrounds.device)

    if isect_state is not None:  # (emit + pair sort + offsets: one native call on the sorted path)
        tiles_per_gauss, isect_ids, flatten_ids, isect_offsets = isect_tiles_finish(isect_state, offsets_for=C)
    else:
        tiles_per_gauss, isect_ids, flatten_ids = isect_tiles(
            means2d, radii, depths, tile_size, tile_width, tile_height,
            packed=packed, n_cameras=C, camera_ids=camera_ids, gaussian_ids=gaussian_ids,
        )
        isect_offsets = isect_offset_encode(isect_ids, C, tile_width, tile_height)

    meta.update(
        {
            "tile_width": tile_width,
            "tile_height": tile_height,
            "tiles_per_gauss": tiles_per_gauss,
            "isect_ids": isect_ids,
            "flatten_ids": flatten_ids,
            "isect_offsets": isect_offsets,
            "width": width,
            "height": height,
            "tile_size": tile_size,
            "n_cameras": C,
        }
    )

    if colors.shape[-1] > channel_chunk:
        n_chunks = (colors.shape[-1] + channel_chunk - 1) // channel_chunk
        render_colors, render_alphas = [], []
        for i in range(n_chunks):
            colors_chunk = colors[..., i * channel_chunk : (i + 1) * channel_chunk]
            backgrounds_chunk = (
                backgrounds[..., i * channel_chunk : (i + 1) * channel_chunk] if backgrounds is not None else None
            )
            rc_, ra_ = rasterize_to_pixels(
                means2d, conics, colors_chunk, opacities, width, height, tile_size, isect_offsets, flatten_ids,
                backgrounds=backgrounds_chunk, packed=packed, absgrad=absgrad, deterministic=deterministic,
            )
            render_colors.append(rc_)
            render_alphas.append(ra_)
        render_colors = torch.cat(render_colors, dim=-1)
        render_alphas = render_alphas[0]  # discard the rest
    else:
        render_colors, render_alphas = rasterize_to_pixels(
            means2d, conics, colors, opacities, width, height, tile_size, isect_offsets, flatten_ids,
            backgrounds=backgrounds, packed=packed, absgrad=absgrad, deterministic=deterministic, prefill=prefill,
        )
    if render_mode in ["ED", "RGB+ED"]:
        # accumulated depth -> expected depth (reference rendering.py:471-477: cat(rc[..., :-1], rc[..., -1:] / ra.clamp(min=1e-10)))
        if render_colors.is_cuda and render_colors.dtype == torch.float32 and render_alphas.dtype == torch.float32:
            render_colors = _ExpectedDepth.apply(render_colors, render_alphas)
        else:
            render_colors = torch.cat(
                [render_colors[..., :-1], render_colors[..., -1:] / render_alphas.clamp(min=1e-10)], dim=-1
            )

    return render_colors, render_alphas, meta
