"""A drop-in for the reference's native module object ``_C`` (``gsplat/cuda/_backend.py:79-141``), on top of the HIP C ABI.

The reference resolves every native op lazily by NAME on that object (``gsplat/cuda/_wrapper.py:9-16``), so its whole
native surface is "an object with these attributes" (``gsplat/cuda/csrc/ext.cpp:3-92``).  ``_C`` below carries the 3DGS
subset -- every name the hot path and the reference's ``tests/test_basic.py`` use -- with the reference's POSITIONAL
torch-tensor signatures and return tuples (``gsplat/cuda/include/bindings.h:34-330``): assigning it in the reference's
``_backend.py`` (``_C = gscodec_studio_amd._c_adapter._C``) lets the reference's own ``_wrapper.py`` -- its autograd
Functions, asserts and ``.contiguous()`` calls -- run unmodified on MI355X.  Out of scope, like everything 2DGS / MCMC /
optimizer in SURVEY section 2: ``*_2dgs``, ``compute_relocation``, ``selective_adam_update`` (AttributeError, as for any
name the module does not have).

What the adapter cannot do better than the signatures allow:
* ``rasterize_to_pixels_bwd`` has no place for the forward's checkpoints, so it runs the plain (unsegmented) backward; the
  package's own ``rasterize_to_pixels`` carries them between its forward and backward and is ~3x faster there;
* ``isect_tiles`` reads ``n_isects`` back inside the call (like ``isect_tiles.cu:200``) instead of overlapping the read-back
  with the colour kernels as ``rasterization()`` does.
Outputs the reference leaves uninitialised (``torch::empty``) are uninitialised here too.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
from torch import Tensor

from . import _backend as B
from ._wrapper import _device_of, _require_gpu, _stream


def _c(t: Optional[Tensor]) -> Optional[Tensor]:
    return None if t is None else t.contiguous()


class _CameraModelType(int):
    """``CameraModelType`` of ext.cpp:4-8: int-valued members with ``.name`` / ``.value``."""

    def __new__(cls, value: int, name: str):
        o = int.__new__(cls, value)
        o._name = name
        return o

    @property
    def name(self) -> str:
        return self._name

    @property
    def value(self) -> int:
        return int(self)

    def __repr__(self) -> str:
        return f"CameraModelType.{self._name}"


class _CameraModelTypeEnum:
    PINHOLE = _CameraModelType(0, "PINHOLE")
    ORTHO = _CameraModelType(1, "ORTHO")
    FISHEYE = _CameraModelType(2, "FISHEYE")


class HipBackend:
    """The attribute set of the reference's pybind module for the 3DGS path; every method is a few calls into the C ABI."""

    CameraModelType = _CameraModelTypeEnum
    PINHOLE, ORTHO, FISHEYE = _CameraModelTypeEnum.PINHOLE, _CameraModelTypeEnum.ORTHO, _CameraModelTypeEnum.FISHEYE

    # ---- spherical harmonics (bindings.h:217-232; compute_sh_fwd.cu:40-72, compute_sh_bwd.cu:53-95)
    @staticmethod
    def compute_sh_fwd(degrees_to_use: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor]) -> Tensor:
        _require_gpu(coeffs, "compute_sh_fwd")
        dirs, coeffs, masks = _c(dirs), _c(coeffs), _c(masks)
        K, n = coeffs.shape[-2], dirs.numel() // 3
        colors = torch.empty_like(dirs)
        m8 = masks.view(torch.uint8) if masks is not None else None
        with _device_of(dirs):
            B.call("gs_sh_fwd", 1, n, K, int(degrees_to_use), B.ptr(dirs), B.ptr(coeffs), 0, B.ptr(m8), B.ptr(colors), _stream(dirs))
        return colors

    @staticmethod
    def compute_sh_bwd(K: int, degrees_to_use: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor], v_colors: Tensor,
                       compute_v_dirs: bool):
        _require_gpu(coeffs, "compute_sh_bwd")
        dirs, coeffs, masks, v_colors = _c(dirs), _c(coeffs), _c(masks), _c(v_colors)
        n = dirs.numel() // 3
        v_coeffs = torch.empty_like(coeffs)  # every row is written (zeros for masked elements / inactive bands)
        v_dirs = torch.empty_like(dirs) if compute_v_dirs else None
        m8 = masks.view(torch.uint8) if masks is not None else None
        with _device_of(dirs):
            B.call("gs_sh_bwd", 1, n, int(K), int(degrees_to_use), B.ptr(dirs), B.ptr(coeffs), 0, B.ptr(m8), B.ptr(v_colors),
                   B.ptr(v_coeffs), B.ptr(v_dirs), _stream(dirs))
        return v_coeffs, v_dirs

    # ---- quat / scale -> covariance / precision (bindings.h:41-55)
    @staticmethod
    def quat_scale_to_covar_preci_fwd(quats: Tensor, scales: Tensor, compute_covar: bool, compute_preci: bool, triu: bool):
        _require_gpu(quats, "quat_scale_to_covar_preci_fwd")
        quats, scales = _c(quats), _c(scales)
        N = quats.shape[0]
        shape = (N, 6) if triu else (N, 3, 3)
        covars = torch.empty(shape, dtype=torch.float32, device=quats.device) if compute_covar else None
        precis = torch.empty(shape, dtype=torch.float32, device=quats.device) if compute_preci else None
        with _device_of(quats):
            B.call("gs_quat_scale_to_covar_preci_fwd", N, B.ptr(quats), B.ptr(scales), int(triu), B.ptr(covars), B.ptr(precis),
                   _stream(quats))
        return covars, precis

    @staticmethod
    def quat_scale_to_covar_preci_bwd(quats: Tensor, scales: Tensor, v_covars: Optional[Tensor], v_precis: Optional[Tensor],
                                      triu: bool):
        _require_gpu(quats, "quat_scale_to_covar_preci_bwd")
        quats, scales, v_covars, v_precis = _c(quats), _c(scales), _c(v_covars), _c(v_precis)
        v_quats, v_scales = torch.empty_like(quats), torch.empty_like(scales)
        with _device_of(quats):
            B.call("gs_quat_scale_to_covar_preci_bwd", quats.shape[0], B.ptr(quats), B.ptr(scales), int(triu), B.ptr(v_covars),
                   B.ptr(v_precis), B.ptr(v_quats), B.ptr(v_scales), _stream(quats))
        return v_quats, v_scales

    # ---- proj / world_to_cam (bindings.h:57-96)
    @staticmethod
    def proj_fwd(means: Tensor, covars: Tensor, Ks: Tensor, width: int, height: int, camera_model):
        _require_gpu(means, "proj_fwd")
        means, covars, Ks = _c(means), _c(covars), _c(Ks)
        C, N = means.shape[0], means.shape[1]
        means2d = torch.empty((C, N, 2), dtype=torch.float32, device=means.device)
        covars2d = torch.empty((C, N, 2, 2), dtype=torch.float32, device=means.device)
        with _device_of(means):
            B.call("gs_proj_fwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(Ks), int(width), int(height), int(camera_model),
                   B.ptr(means2d), B.ptr(covars2d), _stream(means))
        return means2d, covars2d

    @staticmethod
    def proj_bwd(means: Tensor, covars: Tensor, Ks: Tensor, width: int, height: int, camera_model, v_means2d: Tensor,
                 v_covars2d: Tensor):
        _require_gpu(means, "proj_bwd")
        means, covars, Ks, v_means2d, v_covars2d = _c(means), _c(covars), _c(Ks), _c(v_means2d), _c(v_covars2d)
        C, N = means.shape[0], means.shape[1]
        v_means, v_covars = torch.empty_like(means), torch.empty_like(covars)
        with _device_of(means):
            B.call("gs_proj_bwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(Ks), int(width), int(height), int(camera_model),
                   B.ptr(v_means2d), B.ptr(v_covars2d), B.ptr(v_means), B.ptr(v_covars), _stream(means))
        return v_means, v_covars

    @staticmethod
    def world_to_cam_fwd(means: Tensor, covars: Tensor, viewmats: Tensor):
        _require_gpu(means, "world_to_cam_fwd")
        means, covars, viewmats = _c(means), _c(covars), _c(viewmats)
        C, N = viewmats.shape[0], means.shape[0]
        means_c = torch.empty((C, N, 3), dtype=torch.float32, device=means.device)
        covars_c = torch.empty((C, N, 3, 3), dtype=torch.float32, device=means.device)
        with _device_of(means):
            B.call("gs_world_to_cam_fwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(viewmats), B.ptr(means_c), B.ptr(covars_c),
                   _stream(means))
        return means_c, covars_c

    @staticmethod
    def world_to_cam_bwd(means: Tensor, covars: Tensor, viewmats: Tensor, v_means_c: Optional[Tensor],
                         v_covars_c: Optional[Tensor], means_requires_grad: bool, covars_requires_grad: bool,
                         viewmats_requires_grad: bool):
        _require_gpu(means, "world_to_cam_bwd")
        means, covars, viewmats, v_means_c, v_covars_c = _c(means), _c(covars), _c(viewmats), _c(v_means_c), _c(v_covars_c)
        C, N = viewmats.shape[0], means.shape[0]
        v_means = torch.empty_like(means) if means_requires_grad else None
        v_covars = torch.empty_like(covars) if covars_requires_grad else None
        v_viewmats = torch.zeros_like(viewmats) if viewmats_requires_grad else None
        with _device_of(means):
            B.call("gs_world_to_cam_bwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(viewmats), B.ptr(v_means_c), B.ptr(v_covars_c),
                   B.ptr(v_means), B.ptr(v_covars), B.ptr(v_viewmats), _stream(means))
        return v_means, v_covars, v_viewmats

    # ---- fully fused projection (bindings.h:98-151; fully_fused_projection_fwd.cu:198-275, _bwd.cu:265-372)
    @staticmethod
    def fully_fused_projection_fwd(means, covars, quats, scales, viewmats, Ks, image_width, image_height, eps2d, near_plane,
                                   far_plane, radius_clip, calc_compensations, camera_model):
        _require_gpu(means, "fully_fused_projection_fwd")
        means, covars, quats, scales, viewmats, Ks = _c(means), _c(covars), _c(quats), _c(scales), _c(viewmats), _c(Ks)
        C, N, dev = viewmats.shape[0], means.shape[0], means.device
        radii = torch.empty((C, N), dtype=torch.int32, device=dev)
        means2d = torch.empty((C, N, 2), dtype=torch.float32, device=dev)
        depths = torch.empty((C, N), dtype=torch.float32, device=dev)
        conics = torch.empty((C, N, 3), dtype=torch.float32, device=dev)
        comps = torch.zeros((C, N), dtype=torch.float32, device=dev) if calc_compensations else None  # (fwd.cu:242-245)
        with _device_of(means):
            B.call("gs_projection_fwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(quats), B.ptr(scales), B.ptr(viewmats),
                   B.ptr(Ks), int(image_width), int(image_height), float(eps2d), float(near_plane), float(far_plane),
                   float(radius_clip), int(camera_model), B.ptr(radii), B.ptr(means2d), B.ptr(depths), B.ptr(conics),
                   B.ptr(comps), _stream(means))
        return radii, means2d, depths, conics, comps

    @staticmethod
    def fully_fused_projection_bwd(means, covars, quats, scales, viewmats, Ks, image_width, image_height, eps2d, camera_model,
                                   radii, conics, compensations, v_means2d, v_depths, v_conics, v_compensations,
                                   viewmats_requires_grad):
        _require_gpu(means, "fully_fused_projection_bwd")
        means, covars, quats, scales, viewmats, Ks = _c(means), _c(covars), _c(quats), _c(scales), _c(viewmats), _c(Ks)
        radii, conics, compensations = _c(radii), _c(conics), _c(compensations)
        v_means2d, v_depths, v_conics, v_compensations = _c(v_means2d), _c(v_depths), _c(v_conics), _c(v_compensations)
        C, N = viewmats.shape[0], means.shape[0]
        v_means = torch.empty_like(means)  # every row is written by the kernel: no zero fill (the reference's are zeros + atomics)
        v_covars = torch.empty_like(covars) if covars is not None else None
        v_quats = torch.empty_like(quats) if covars is None else None
        v_scales = torch.empty_like(scales) if covars is None else None
        v_viewmats = torch.zeros_like(viewmats) if viewmats_requires_grad else None
        with _device_of(means):
            B.call("gs_projection_bwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(quats), B.ptr(scales), B.ptr(viewmats),
                   B.ptr(Ks), int(image_width), int(image_height), float(eps2d), int(camera_model), B.ptr(radii), B.ptr(conics),
                   B.ptr(compensations), B.ptr(v_means2d), B.ptr(v_depths), B.ptr(v_conics), B.ptr(v_compensations),
                   B.ptr(v_means), B.ptr(v_covars), B.ptr(v_quats), B.ptr(v_scales), B.ptr(v_viewmats), 2, 3, None, _stream(means))
        return v_means, v_covars, v_quats, v_scales, v_viewmats

    # ---- packed projection (bindings.h:237-293)
    @staticmethod
    def fully_fused_projection_packed_fwd(means, covars, quats, scales, viewmats, Ks, image_width, image_height, eps2d,
                                          near_plane, far_plane, radius_clip, calc_compensations, camera_model):
        _require_gpu(means, "fully_fused_projection_packed_fwd")
        means, covars, quats, scales, viewmats, Ks = _c(means), _c(covars), _c(quats), _c(scales), _c(viewmats), _c(Ks)
        C, N, dev = viewmats.shape[0], means.shape[0], means.device
        nblocks = (N + 255) // 256
        st = _stream(means)
        common = (C, N, B.ptr(means), B.ptr(covars), B.ptr(quats), B.ptr(scales), B.ptr(viewmats), B.ptr(Ks), int(image_width),
                  int(image_height), float(eps2d), float(near_plane), float(far_plane), float(radius_clip), int(camera_model))
        with _device_of(means):
            nnz = 0
            if C * N > 0:
                block_cnts = torch.empty(C * nblocks, dtype=torch.int32, device=dev)
                B.call("gs_projection_packed_count", *common, B.ptr(block_cnts), st)
                block_accum = torch.empty_like(block_cnts)
                sb = B.query("gs_cumsum_scratch_bytes", C * nblocks)
                scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
                B.call("gs_cumsum_i32_i32", C * nblocks, B.ptr(block_cnts), B.ptr(block_accum), B.ptr(scratch), sb, st)
                nnz = int(block_accum[-1].item())  # (packed_fwd.cu:335)
            indptr = torch.zeros(C + 1, dtype=torch.int32, device=dev)
            camera_ids = torch.empty(nnz, dtype=torch.int64, device=dev)
            gaussian_ids = torch.empty(nnz, dtype=torch.int64, device=dev)
            radii = torch.empty(nnz, dtype=torch.int32, device=dev)
            means2d = torch.empty((nnz, 2), dtype=torch.float32, device=dev)
            depths = torch.empty(nnz, dtype=torch.float32, device=dev)
            conics = torch.empty((nnz, 3), dtype=torch.float32, device=dev)
            comps = torch.zeros(nnz, dtype=torch.float32, device=dev) if calc_compensations else None
            if nnz > 0:
                B.call("gs_projection_packed_fill", *common, B.ptr(block_accum), B.ptr(indptr), B.ptr(camera_ids), B.ptr(gaussian_ids),
                       B.ptr(radii), B.ptr(means2d), B.ptr(depths), B.ptr(conics), B.ptr(comps), st)
        return indptr, camera_ids, gaussian_ids, radii, means2d, depths, conics, comps

    @staticmethod
    def fully_fused_projection_packed_bwd(means, covars, quats, scales, viewmats, Ks, image_width, image_height, eps2d,
                                          camera_model, camera_ids, gaussian_ids, conics, compensations, v_means2d, v_depths,
                                          v_conics, v_compensations, viewmats_requires_grad, sparse_grad):
        _require_gpu(means, "fully_fused_projection_packed_bwd")
        means, covars, quats, scales, viewmats, Ks = _c(means), _c(covars), _c(quats), _c(scales), _c(viewmats), _c(Ks)
        camera_ids, gaussian_ids, conics, compensations = _c(camera_ids), _c(gaussian_ids), _c(conics), _c(compensations)
        v_means2d, v_depths, v_conics, v_compensations = _c(v_means2d), _c(v_depths), _c(v_conics), _c(v_compensations)
        C, N, nnz, dev = viewmats.shape[0], means.shape[0], camera_ids.shape[0], means.device

        def buf(like, width):
            if like is None:
                return None
            return (torch.empty((nnz, width), dtype=torch.float32, device=dev) if sparse_grad
                    else torch.zeros((N, width), dtype=torch.float32, device=dev))

        v_means = buf(means, 3)
        v_covars = buf(covars, 6)
        v_quats = buf(quats, 4) if covars is None else None
        v_scales = buf(scales, 3) if covars is None else None
        v_viewmats = torch.zeros_like(viewmats) if viewmats_requires_grad else None
        with _device_of(means):
            B.call("gs_projection_packed_bwd", C, N, nnz, B.ptr(means), B.ptr(covars), B.ptr(quats), B.ptr(scales), B.ptr(viewmats),
                   B.ptr(Ks), int(image_width), int(image_height), float(eps2d), int(camera_model), B.ptr(camera_ids),
                   B.ptr(gaussian_ids), B.ptr(conics), B.ptr(compensations), B.ptr(v_means2d), B.ptr(v_depths), B.ptr(v_conics),
                   B.ptr(v_compensations), int(bool(sparse_grad)), B.ptr(v_means), B.ptr(v_covars), B.ptr(v_quats), B.ptr(v_scales),
                   B.ptr(v_viewmats), _stream(means))
        return v_means, v_covars, v_quats, v_scales, v_viewmats

    # ---- tile binning (bindings.h:153-173; isect_tiles.cu:106-306, 356-389)
    @staticmethod
    def isect_tiles(means2d, radii, depths, camera_ids, gaussian_ids, C, tile_size, tile_width, tile_height, sort,
                    double_buffer):
        from ._wrapper import isect_tiles as _isect

        packed = camera_ids is not None
        return _isect(means2d, radii, depths, int(tile_size), int(tile_width), int(tile_height), sort=bool(sort), packed=packed,
                      n_cameras=int(C), camera_ids=camera_ids, gaussian_ids=gaussian_ids)

    @staticmethod
    def isect_offset_encode(isect_ids: Tensor, C: int, tile_width: int, tile_height: int) -> Tensor:
        _require_gpu(isect_ids, "isect_offset_encode")
        isect_ids = _c(isect_ids)
        n_tiles = tile_width * tile_height
        tile_n_bits = int(math.floor(math.log2(n_tiles))) + 1 if n_tiles > 0 else 1
        offsets = torch.empty((C, tile_height, tile_width), dtype=torch.int32, device=isect_ids.device)
        with _device_of(isect_ids):
            B.call("gs_isect_offset_encode", isect_ids.shape[0], B.ptr(isect_ids), int(C), n_tiles, tile_n_bits, B.ptr(offsets),
                   _stream(isect_ids))
        return offsets

    # ---- compositing (bindings.h:175-215; rasterize_to_pixels_fwd.cu:187-352, _bwd.cu:279-489)
    @staticmethod
    def rasterize_to_pixels_fwd(means2d, conics, colors, opacities, backgrounds, masks, image_width, image_height, tile_size,
                                tile_offsets, flatten_ids):
        _require_gpu(means2d, "rasterize_to_pixels_fwd")
        means2d, conics, colors, opacities = _c(means2d), _c(conics), _c(colors), _c(opacities)
        backgrounds, masks, tile_offsets, flatten_ids = _c(backgrounds), _c(masks), _c(tile_offsets), _c(flatten_ids)
        C, tile_height, tile_width = tile_offsets.shape
        channels, dev = colors.shape[-1], means2d.device
        renders = torch.empty((C, image_height, image_width, channels), dtype=torch.float32, device=dev)
        alphas = torch.empty((C, image_height, image_width, 1), dtype=torch.float32, device=dev)
        last_ids = torch.empty((C, image_height, image_width), dtype=torch.int32, device=dev)
        m8 = masks.view(torch.uint8) if masks is not None else None
        with _device_of(means2d):
            B.call("gs_rasterize_fwd", C, opacities.numel(), flatten_ids.shape[0], channels, B.ptr(means2d), B.ptr(conics),
                   B.ptr(colors), B.ptr(opacities), None, B.ptr(backgrounds), B.ptr(m8), int(image_width), int(image_height),
                   int(tile_size), tile_width, tile_height, B.ptr(tile_offsets), B.ptr(flatten_ids), B.ptr(renders), B.ptr(alphas),
                   B.ptr(last_ids), None, None, None, 0, _stream(means2d))
        return renders, alphas, last_ids

    @staticmethod
    def rasterize_to_pixels_bwd(means2d, conics, colors, opacities, backgrounds, masks, image_width, image_height, tile_size,
                                tile_offsets, flatten_ids, render_alphas, last_ids, v_render_colors, v_render_alphas, absgrad):
        _require_gpu(means2d, "rasterize_to_pixels_bwd")
        means2d, conics, colors, opacities = _c(means2d), _c(conics), _c(colors), _c(opacities)
        backgrounds, masks, tile_offsets, flatten_ids = _c(backgrounds), _c(masks), _c(tile_offsets), _c(flatten_ids)
        render_alphas, last_ids, v_render_colors, v_render_alphas = _c(render_alphas), _c(last_ids), _c(v_render_colors), _c(v_render_alphas)
        C, tile_height, tile_width = tile_offsets.shape
        channels = colors.shape[-1]
        v_means2d, v_conics = torch.zeros_like(means2d), torch.zeros_like(conics)
        v_colors, v_opacities = torch.zeros_like(colors), torch.zeros_like(opacities)
        v_means2d_abs = torch.zeros_like(means2d) if absgrad else None
        m8 = masks.view(torch.uint8) if masks is not None else None
        with _device_of(means2d):
            B.call("gs_rasterize_bwd", C, opacities.numel(), flatten_ids.shape[0], channels, B.ptr(means2d), B.ptr(conics),
                   B.ptr(colors), B.ptr(opacities), None, B.ptr(backgrounds), B.ptr(m8), int(image_width), int(image_height),
                   int(tile_size), tile_width, tile_height, B.ptr(tile_offsets), B.ptr(flatten_ids), None, B.ptr(render_alphas),
                   B.ptr(last_ids), B.ptr(v_render_colors), B.ptr(v_render_alphas), channels, 1, B.ptr(v_means2d_abs),
                   B.ptr(v_means2d), B.ptr(v_conics), B.ptr(v_colors), B.ptr(v_opacities), 0, None, None, None, _stream(means2d))
        return v_means2d_abs, v_means2d, v_conics, v_colors, v_opacities

    @staticmethod
    def rasterize_to_indices_in_range(range_start, range_end, transmittances, means2d, conics, opacities, image_width,
                                      image_height, tile_size, tile_offsets, flatten_ids):
        from ._wrapper import rasterize_to_indices_in_range as _r2i

        gaussian_ids, pixel_ids, camera_ids = _r2i(range_start, range_end, transmittances, means2d, conics, opacities, image_width,
                                                   image_height, tile_size, tile_offsets, flatten_ids)
        # the native function returns the pixel id INCLUDING the camera (the reference's Python splits it, _wrapper.py:636-643)
        return gaussian_ids, camera_ids * (image_width * image_height) + pixel_ids


_C = HipBackend()
