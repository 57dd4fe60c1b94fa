"""Operator layer: same public functions as the reference's ``gsplat/cuda/_wrapper.py``
(lines 47-643, 775-1256), dispatching to the HIP C ABI instead of a pybind module.

Each function keeps the reference's name, argument meaning, defaults, return shapes /
dtypes and error behaviour (shape asserts in Python, ``RuntimeError`` from native
failures).  PyTorch owns memory, autograd and streams; every kernel is a call through
``_backend.call`` with raw device pointers and the current HIP stream.

What is different by design:
* gradient buffers that a kernel fully overwrites are ``torch.empty`` (no zero-fill pass);
* SH coefficients shared by all cameras are never expanded to ``[C,N,K,3]``;
* the tile-intersection prefix sum and the 64-bit radix sort are this library's own
  kernels (the reference uses ``torch::cumsum`` and ``cub::DeviceRadixSort``).
"""
from __future__ import annotations

import ctypes
import math
import os
import struct
from typing import Optional, Tuple

import torch
from torch import Tensor
from typing_extensions import Literal

from . import _backend as B

_CAMERA_MODELS = {"pinhole": 0, "ortho": 1, "fisheye": 2}
# splat rows (include/gsplat_hip.h): one 64-byte row of 16 floats per projected splat
ROW, ROW_MEAN2D, ROW_CONIC, ROW_OPACITY, ROW_COLOR, ROW_DEPTH, ROW_RADIUS, ROW_COMP = 16, 0, 2, 5, 6, 9, 10, 11


def _require_gpu(t: Tensor, what: str) -> None:
    if not t.is_cuda:
        raise RuntimeError(
            f"{what}: expected a GPU tensor (got device {t.device}). The HIP path has no CPU "
            "fallback; the CPU restatement lives in oracle/ and is test infrastructure only."
        )


def _f32c(t: Optional[Tensor]) -> Optional[Tensor]:
    if t is None:
        return None
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32 tensor, got {t.dtype}")
    return t.contiguous()


def _stream(t: Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


class _device_of:
    """``with _device_of(t):`` -- make t's device current (GSPLAT_DEVICE_GUARD equivalent)."""

    def __init__(self, t: Tensor):
        self._g = torch.cuda.device(t.device)

    def __enter__(self):
        return self._g.__enter__()

    def __exit__(self, *a):
        return self._g.__exit__(*a)


# ---------------------------------------------------------------------------
# spherical harmonics  (reference _wrapper.py:47-73, 1226-1256)
# ---------------------------------------------------------------------------
def spherical_harmonics(
    degrees_to_use: int,
    dirs: Tensor,  # [..., 3]
    coeffs: Tensor,  # [..., K, 3]
    masks: Optional[Tensor] = None,
) -> Tensor:
    """Computes spherical harmonics.

    Args:
        degrees_to_use: The degree to be used.
        dirs: Directions. [..., 3]
        coeffs: Coefficients. [..., K, 3]
        masks: Optional boolen masks to skip some computation. [...,] Default: None.

    Returns:
        Spherical harmonics. [..., 3]
    """
    assert (degrees_to_use + 1) ** 2 <= coeffs.shape[-2], coeffs.shape
    assert dirs.shape[:-1] == coeffs.shape[:-2], (dirs.shape, coeffs.shape)
    assert dirs.shape[-1] == 3, dirs.shape
    assert coeffs.shape[-1] == 3, coeffs.shape
    if masks is not None:
        assert masks.shape == dirs.shape[:-1], masks.shape
        masks = masks.contiguous()
    return _SphericalHarmonics.apply(degrees_to_use, dirs.contiguous(), coeffs.contiguous(), masks, False)


def spherical_harmonics_shared(
    degrees_to_use: int,
    dirs: Tensor,  # [C, N, 3]
    coeffs: Tensor,  # [N, K, 3]  (shared by all cameras)
    masks: Optional[Tensor] = None,  # [C, N]
) -> Tensor:
    """SH evaluation for coefficients shared by all C cameras.

    Equivalent to ``spherical_harmonics(deg, dirs, coeffs.expand(C, -1, -1, -1), masks)``
    (what the reference's ``rasterization`` does, rendering.py:386-390) but never
    materialises the ``[C,N,K,3]`` coefficient copy (reference _wrapper.py:72) nor the
    ``[C,N,K,3]`` gradient: the kernel indexes ``[N,K,3]`` directly and the backward
    sums over cameras in registers.
    """
    C, N = dirs.shape[0], dirs.shape[1]
    assert dirs.shape == (C, N, 3), dirs.shape
    assert coeffs.dim() == 3 and coeffs.shape[0] == N and coeffs.shape[2] == 3, coeffs.shape
    assert (degrees_to_use + 1) ** 2 <= coeffs.shape[-2], coeffs.shape
    if masks is not None:
        assert masks.shape == (C, N), masks.shape
        masks = masks.contiguous()
    return _SphericalHarmonics.apply(degrees_to_use, dirs.contiguous(), coeffs.contiguous(), masks, True)


def _row_strided(t: Tensor, width: int):
    """(tensor, row_stride) for a [..., width] gradient that is either contiguous or a column
    slice of a wider row-major buffer (e.g. the packed [n_elems,16] compositing gradients): such
    views are consumed in place by the kernels instead of being copied by ``.contiguous()``."""
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32 tensor, got {t.dtype}")
    if t.is_contiguous():
        return t, width
    if t.dim() >= 2 and t.shape[-1] == width and t.stride(-1) == 1:
        rs = t.stride(-2)
        expect = rs
        ok = True
        for d in range(t.dim() - 2, -1, -1):
            if t.shape[d] != 1 and t.stride(d) != expect:
                ok = False
                break
            expect *= t.shape[d]
        if ok and rs >= width:
            return t, rs
    return t.contiguous(), width


def _elem_strided(t: Tensor):
    """(tensor, element_stride) for a [C, N] gradient that is contiguous or one column of a wider row-major buffer
    (the opacity slot of the packed [C*N, 16] compositing gradients); anything else is copied."""
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32 tensor, got {t.dtype}")
    if t.is_contiguous():
        return t, 1
    if t.dim() == 1 and t.stride(0) >= 1:
        return t, t.stride(0)
    if t.dim() == 2 and t.stride(1) >= 1 and (t.shape[0] == 1 or t.stride(0) == t.shape[1] * t.stride(1)):
        return t, t.stride(1)
    return t.contiguous(), 1


def _pixel_strided(t: Tensor):
    """(tensor, pixel stride, channel stride) of an image gradient [C,H,W,D] whose pixels are laid out uniformly: a dense
    tensor -> (D, 1); the expanded scalar autograd produces for ``sum(render)`` -> (0, 0), read in place by the kernels
    instead of being materialised (25 MB at 1080p); anything else is copied."""
    if t.dtype != torch.float32:
        raise RuntimeError(f"expected float32 tensor, got {t.dtype}")
    if t.is_contiguous():
        return t, t.shape[-1], 1
    C, H, W, D = t.shape
    sp, sc = t.stride(2), t.stride(3)
    if sp >= 0 and sc >= 0 and t.stride(1) == W * sp and t.stride(0) == H * W * sp and (sp == 0 or sp >= D * max(sc, 1)):
        return t, sp, sc
    return t.contiguous(), D, 1


def _rows_table(parts, n_rows: int, row_index=None):
    """ctypes tables for gs_rows_*: parts = [(tensor | None, width[, indexed])].  An indexed part pairs wire row r with
    ITS row row_index[r] (any number of rows); the others have exactly n_rows rows."""
    import ctypes

    n = len(parts)
    ptrs, widths, strides = (ctypes.c_void_p * n)(), (ctypes.c_int32 * n)(), (ctypes.c_int64 * n)()
    flags, keep = (ctypes.c_int32 * n)(), []
    for k, part in enumerate(parts):
        t, w = part[0], part[1]
        indexed = bool(part[2]) if len(part) > 2 else False
        widths[k], strides[k], ptrs[k], flags[k] = w, w, None, int(indexed)
        if t is None:
            continue
        if t.dtype == torch.int32:
            t = t.view(torch.float32)
        if not indexed:
            assert t.numel() == n_rows * w, (tuple(t.shape), n_rows, w)
        else:
            assert row_index is not None and t.numel() % w == 0
        if w == 1:
            t, rs = _elem_strided(t if t.dim() == 2 else t.reshape(1, -1))
        else:
            t, rs = _row_strided(t, w)
        keep.append(t)
        ptrs[k], strides[k] = t.data_ptr(), rs
    return n, ptrs, widths, strides, flags, keep


def _index_arg(row_index):
    """(pointer, element stride) of an int32 index list: a contiguous [n] tensor or one column of a row-major buffer."""
    if row_index is None:
        return None, 1
    assert row_index.dtype == torch.int32 and row_index.dim() == 1
    return row_index.data_ptr(), int(row_index.stride(0)) if row_index.numel() > 1 else 1


def rows_pack(parts, n_rows: int, like: Tensor, row_index: Optional[Tensor] = None) -> Tensor:
    """Gather column blocks into wire rows: ``parts`` = [(tensor [..., w] fp32 / int32 (bit pattern) or None = zeros, w
    [, indexed])]; returns [n_rows, sum(w)] fp32.  Column views of wider row-major buffers are read in place; with
    ``row_index`` (int32 [n_rows]) the indexed parts are read at row ``row_index[r]`` (gs_rows_pack[_indexed])."""
    _require_gpu(like, "rows_pack")
    import ctypes

    n, ptrs, widths, strides, flags, keep = _rows_table(parts, n_rows, row_index)
    wire = torch.empty((n_rows, sum(p[1] for p in parts)), dtype=torch.float32, device=like.device)
    ip, istride = _index_arg(row_index)
    with _device_of(like):
        if row_index is None:
            B.call("gs_rows_pack", n_rows, n, ctypes.addressof(ptrs), ctypes.addressof(widths), ctypes.addressof(strides),
                   B.ptr(wire), _stream(like))
        else:
            B.call("gs_rows_pack_indexed", n_rows, n, ctypes.addressof(ptrs), ctypes.addressof(widths), ctypes.addressof(strides),
                   ctypes.addressof(flags), ip, istride, B.ptr(wire), _stream(like))
    return wire


def rows_unpack(wire: Tensor, parts, row_index: Optional[Tensor] = None) -> None:
    """Scatter wire rows [n_rows, sum(w)] into ``parts`` = [(contiguous destination tensor or None = skipped, w[, indexed])];
    indexed parts are written at row ``row_index[r]`` (``row_index`` may be an int32 column view of the wire itself)."""
    _require_gpu(wire, "rows_unpack")
    assert wire.is_contiguous() and wire.dtype == torch.float32
    for part in parts:
        assert part[0] is None or part[0].is_contiguous()
    import ctypes

    n_rows = wire.shape[0]
    n, ptrs, widths, strides, flags, keep = _rows_table(parts, n_rows, row_index)
    assert sum(p[1] for p in parts) == wire.shape[1]
    ip, istride = _index_arg(row_index)
    with _device_of(wire):
        if row_index is None:
            B.call("gs_rows_unpack", n_rows, n, ctypes.addressof(ptrs), ctypes.addressof(widths), ctypes.addressof(strides),
                   B.ptr(wire), _stream(wire))
        else:
            B.call("gs_rows_unpack_indexed", n_rows, n, ctypes.addressof(ptrs), ctypes.addressof(widths), ctypes.addressof(strides),
                   ctypes.addressof(flags), ip, istride, B.ptr(wire), _stream(wire))


class _GatherRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, src: Tensor, ids: Tensor) -> Tensor:
        _require_gpu(src, "gather_rows")
        src = _f32c(src)
        ids = ids.contiguous()
        assert ids.dtype == torch.int64 and ids.dim() == 1
        width = int(src.numel() // max(src.shape[0], 1))
        out = torch.empty((ids.shape[0],) + tuple(src.shape[1:]), dtype=torch.float32, device=src.device)
        with _device_of(src):
            B.call("gs_gather_rows_f32", ids.shape[0], width, B.ptr(src), B.ptr(ids), B.ptr(out), _stream(src))
        ctx.save_for_backward(ids)
        ctx.src_shape, ctx.width = tuple(src.shape), width
        return out

    @staticmethod
    def backward(ctx, v_out: Tensor):
        (ids,) = ctx.saved_tensors
        v_out = _f32c(v_out)
        v_src = torch.zeros(ctx.src_shape, dtype=torch.float32, device=v_out.device)
        with _device_of(v_out):
            B.call("gs_scatter_add_rows_f32", ids.shape[0], ctx.width, B.ptr(v_out), B.ptr(ids), B.ptr(v_src), _stream(v_out))
        return v_src, None


def gather_rows(src: Tensor, ids: Tensor) -> Tensor:
    """``src[ids]`` for fp32 ``src`` [N, ...] and int64 ``ids`` [nnz] with a one-pass atomic backward (the packed pipeline's
    per-splat gathers; torch's indexing backward sorts the ids and costs ~0.45 ms per step at 2.8 M splats)."""
    if not src.is_cuda or src.dtype != torch.float32 or ids.dtype != torch.int64 or ids.dim() != 1:
        return src[ids]
    return _GatherRows.apply(src, ids)


def exchange_compact(radii: Tensor, C_local: int, world: int, cap: int, N_total: int, N_off: int):
    """Lists of the visible rows of ``radii`` [C_total, N] per destination rank (gs_exchange_compact): returns
    (src_index i32 [world * (cap + 1)], hdr i32 [world * (cap + 1), 2], counters i32 [world], stats i32 [2])."""
    _require_gpu(radii, "exchange_compact")
    assert radii.dtype == torch.int32 and radii.is_contiguous() and radii.dim() == 2
    C_total, N = radii.shape
    rows = world * (cap + 1)
    dev = radii.device
    src_index = torch.empty(rows, dtype=torch.int32, device=dev)
    hdr = torch.empty((rows, 2), dtype=torch.int32, device=dev)
    counters = torch.empty(world, dtype=torch.int32, device=dev)
    stats = torch.empty(2, dtype=torch.int32, device=dev)
    with _device_of(radii):
        B.call("gs_exchange_compact", C_total, N, C_local, world, cap, N_total, N_off, B.ptr(radii), B.ptr(src_index), B.ptr(hdr),
               B.ptr(counters), B.ptr(stats), _stream(radii))
    return src_index, hdr, counters, stats


def rows16_gather(n_rows: int, index: Tensor, index_stride: int, src_rows: Tensor, tag: Optional[Tensor] = None) -> Tensor:
    """``out[r] = src_rows[index[r * index_stride]]`` over 64-byte splat rows (zeros for a negative index); ``tag`` [n_rows, 2]
    int32 goes into columns 12 / 13 (gs_rows16_gather).  ``index`` may be a column of another row buffer."""
    _require_gpu(src_rows, "rows16_gather")
    assert index.dtype == torch.int32 and src_rows.dtype == torch.float32 and src_rows.is_contiguous() and src_rows.shape[-1] == ROW
    out = torch.empty((n_rows, ROW), dtype=torch.float32, device=src_rows.device)
    with _device_of(src_rows):
        B.call("gs_rows16_gather", n_rows, index.data_ptr(), index_stride, B.ptr(src_rows), B.ptr(tag), B.ptr(out), _stream(src_rows))
    return out


def rows16_scatter(n_rows: int, index: Tensor, index_stride: int, wire: Tensor, dst_rows: Tensor, radii: Optional[Tensor] = None,
                   depths: Optional[Tensor] = None) -> None:
    """``dst_rows[index[r * index_stride]] = wire[r]`` where the index is >= 0; ``radii`` / ``depths`` receive the rows' columns
    10 / 9 at the same element (gs_rows16_scatter)."""
    _require_gpu(wire, "rows16_scatter")
    assert index.dtype == torch.int32 and wire.is_contiguous() and dst_rows.is_contiguous() and dst_rows.shape[-1] == ROW
    with _device_of(wire):
        B.call("gs_rows16_scatter", n_rows, index.data_ptr(), index_stride, B.ptr(wire), B.ptr(dst_rows), B.ptr(radii), B.ptr(depths),
               _stream(wire))


def spherical_harmonics_view(
    degrees_to_use: int,
    means: Tensor,  # [N, 3]
    campos: Tensor,  # [C, 3] camera centres in world space
    coeffs: Tensor,  # [N, K, 3] shared by all cameras
    radii: Optional[Tensor] = None,  # [C, N] int32: evaluate only where radii > 0
    opacities: Optional[Tensor] = None,  # [N]: also return opacities.repeat(C, 1) (and sum its gradient over cameras)
    rows: Optional[Tensor] = None,  # [C, N, 16] splat rows: the colours are written into columns 6:9 and returned as that view
    coeffs_rest: Optional[Tensor] = None,  # SPLIT rows: ``coeffs`` is the DC band [N, 1, 3], this the higher bands [N, K-1, 3]
):
    """Fused colour evaluation used by ``rasterization``:
    ``clamp_min(spherical_harmonics(deg, means[None] - campos[:, None], coeffs, radii > 0) + 0.5, 0)``
    (reference rendering.py:372-392) in ONE kernel each way -- no ``dirs`` / mask / clamp
    tensors, and the backward writes ``v_means`` (summed over cameras) directly."""
    C, N = campos.shape[0], means.shape[0]
    # campos: [C, 3] camera centres, or the [C, 4, 4] world->camera matrices themselves (centre derived in-kernel)
    assert means.shape == (N, 3) and (campos.shape == (C, 3) or campos.shape == (C, 4, 4)), (means.shape, campos.shape)
    assert coeffs.dim() == 3 and coeffs.shape[0] == N and coeffs.shape[2] == 3, coeffs.shape
    if coeffs_rest is not None:  # the trainer's sh0 / shN parameters as they are (no torch.cat, reference simple_trainer.py:779-786)
        assert coeffs.shape == (N, 1, 3) and coeffs_rest.dim() == 3 and coeffs_rest.shape[0] == N and coeffs_rest.shape[2] == 3, \
            (coeffs.shape, coeffs_rest.shape)
        assert coeffs_rest.shape[1] >= 1, coeffs_rest.shape
        coeffs_rest = coeffs_rest.contiguous()
    assert (degrees_to_use + 1) ** 2 <= coeffs.shape[-2] + (coeffs_rest.shape[1] if coeffs_rest is not None else 0), coeffs.shape
    if radii is not None:
        assert radii.shape == (C, N) and radii.dtype == torch.int32, (radii.shape, radii.dtype)
    if opacities is not None:
        assert opacities.shape == (N,), opacities.shape
    if rows is not None:
        assert rows.shape == (C, N, ROW) and rows.is_contiguous() and rows.dtype == torch.float32, (rows.shape, rows.dtype)
    colors, opac_cn = _SphericalHarmonicsView.apply(degrees_to_use, means.contiguous(), campos.contiguous(), coeffs.contiguous(),
                                                    radii.contiguous() if radii is not None else None,
                                                    opacities.contiguous() if opacities is not None else None, rows, coeffs_rest)
    return colors if opacities is None else (colors, opac_cn)


class _SphericalHarmonicsView(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sh_degree, means, campos, coeffs, radii, opacities, rows=None, coeffs_rest=None):
        _require_gpu(coeffs, "spherical_harmonics_view")
        means, campos, coeffs, coeffs_rest = _f32c(means), _f32c(campos), _f32c(coeffs), _f32c(coeffs_rest)
        C, N, K = campos.shape[0], means.shape[0], coeffs.shape[1] + (coeffs_rest.shape[1] if coeffs_rest is not None else 0)
        # the colours are their own [C,N,3] tensor, or columns 6:9 of the splat rows the projection filled (written through
        # the raw pointer: a view of a buffer that is no differentiable input of this node)
        colors = torch.empty((C, N, 3), dtype=torch.float32, device=means.device) if rows is None else rows[..., ROW_COLOR:ROW_COLOR + 3]
        cstride = 3 if rows is None else ROW
        opacities = _f32c(opacities) if opacities is not None else None
        opac_cn = torch.empty((C, N), dtype=torch.float32, device=means.device) if opacities is not None else None
        with _device_of(means):
            B.call("gs_sh_view_fwd", C, N, K, sh_degree, B.ptr(means), B.ptr(campos), int(campos.dim() == 3), B.ptr(coeffs),
                   B.ptr(coeffs_rest), B.ptr(radii), B.ptr(colors), cstride, B.ptr(opacities), B.ptr(opac_cn), _stream(means))
        ctx.save_for_backward(means, campos, coeffs, radii, colors, coeffs_rest)
        ctx.sh_degree, ctx.cstride = sh_degree, cstride
        ctx.has_opac = opacities is not None
        ctx.set_materialize_grads(False)
        return colors, opac_cn

    @staticmethod
    def backward(ctx, v_colors, v_opac_cn):
        means, campos, coeffs, radii, colors, coeffs_rest = ctx.saved_tensors
        C, N, K = campos.shape[0], means.shape[0], coeffs.shape[1] + (coeffs_rest.shape[1] if coeffs_rest is not None else 0)
        if v_colors is None:
            v_colors = torch.zeros_like(colors)
        v_colors, vstride = _row_strided(v_colors, 3)
        v_coeffs = torch.empty_like(coeffs)
        v_rest = torch.empty_like(coeffs_rest) if coeffs_rest is not None else None
        v_means = torch.empty_like(means) if ctx.needs_input_grad[1] else None
        v_opac = ostride = None
        if ctx.has_opac and ctx.needs_input_grad[5]:
            v_opac = torch.empty((N,), dtype=torch.float32, device=means.device)
            if v_opac_cn is None:
                v_opac.zero_()
            else:
                v_opac_cn, ostride = _elem_strided(v_opac_cn)
        with _device_of(means):
            B.call("gs_sh_view_bwd", C, N, K, ctx.sh_degree, B.ptr(means), B.ptr(campos), int(campos.dim() == 3), B.ptr(coeffs),
                   B.ptr(coeffs_rest), B.ptr(radii),
                   B.ptr(colors), ctx.cstride, B.ptr(v_colors), vstride, B.ptr(v_coeffs), B.ptr(v_rest), B.ptr(v_means),
                   B.ptr(v_opac_cn) if ostride is not None else None, ostride or 0, B.ptr(v_opac) if ostride is not None else None,
                   0, _stream(means))
        if not ctx.needs_input_grad[3]:
            v_coeffs = None
        # campos (camera poses) gets no gradient on this path; rasterization() takes the unfused
        # route when viewmats require grad.
        return None, v_means, None, v_coeffs, None, v_opac, None, v_rest


class _SphericalHarmonics(torch.autograd.Function):
    @staticmethod
    def forward(ctx, sh_degree: int, dirs: Tensor, coeffs: Tensor, masks: Optional[Tensor],
                shared: bool = False) -> Tensor:
        _require_gpu(coeffs, "spherical_harmonics")
        dirs, coeffs = _f32c(dirs), _f32c(coeffs)
        K = coeffs.shape[-2]
        if shared:
            C, N = dirs.shape[0], dirs.shape[1]
        else:
            C, N = 1, dirs.numel() // 3
        colors = torch.empty_like(dirs)
        m8 = masks.view(torch.uint8) if masks is not None else None
        with _device_of(dirs):
            B.call("gs_sh_fwd", C, N, K, sh_degree, B.ptr(dirs), B.ptr(coeffs), int(shared), B.ptr(m8),
                   B.ptr(colors), _stream(dirs))
        ctx.save_for_backward(dirs, coeffs, masks)
        ctx.sh_degree = sh_degree
        ctx.layout = (C, N, shared, K)
        return colors

    @staticmethod
    def backward(ctx, v_colors: Tensor):
        dirs, coeffs, masks = ctx.saved_tensors
        C, N, shared, K = ctx.layout
        compute_v_dirs = ctx.needs_input_grad[1]
        v_colors = _f32c(v_colors)
        # every row is written by the kernel (zeros for masked / inactive bands): no zero fill
        v_coeffs = torch.empty_like(coeffs)
        v_dirs = torch.empty_like(dirs) if compute_v_dirs else None
        m8 = masks.view(torch.uint8) if masks is not None else None
        with _device_of(dirs):
            B.call("gs_sh_bwd", C, N, K, ctx.sh_degree, B.ptr(dirs), B.ptr(coeffs), int(shared), B.ptr(m8),
                   B.ptr(v_colors), B.ptr(v_coeffs), B.ptr(v_dirs), _stream(dirs))
        if not ctx.needs_input_grad[2]:
            v_coeffs = None
        return None, v_dirs, v_coeffs, None, None


# ---------------------------------------------------------------------------
# unfused public ops: world_to_cam, proj / persp_proj, rasterize_to_indices_in_range
# (reference _wrapper.py:118-200, 571-643, 709-772)
# ---------------------------------------------------------------------------
_CAMERA_MODELS = {"pinhole": 0, "ortho": 1, "fisheye": 2}


def world_to_cam(
    means: Tensor,  # [N, 3]
    covars: Tensor,  # [N, 3, 3]
    viewmats: Tensor,  # [C, 4, 4]
) -> Tuple[Tensor, Tensor]:
    """Transforms Gaussians from world to camera coordinate system.

    Returns (means_c [C, N, 3], covars_c [C, N, 3, 3])."""
    C = viewmats.size(0)
    N = means.size(0)
    assert means.size() == (N, 3), means.size()
    assert covars.size() == (N, 3, 3), covars.size()
    assert viewmats.size() == (C, 4, 4), viewmats.size()
    return _WorldToCam.apply(means.contiguous(), covars.contiguous(), viewmats.contiguous())


class _WorldToCam(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, covars, viewmats):
        _require_gpu(means, "world_to_cam")
        means, covars, viewmats = _f32c(means), _f32c(covars), _f32c(viewmats)
        C, N = viewmats.shape[0], means.shape[0]
        means_c = torch.empty((C, N, 3), device=means.device)
        covars_c = torch.empty((C, N, 3, 3), device=means.device)
        with _device_of(means):
            B.call("gs_world_to_cam_fwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(viewmats), B.ptr(means_c),
                   B.ptr(covars_c), _stream(means))
        ctx.save_for_backward(means, covars, viewmats)
        return means_c, covars_c

    @staticmethod
    def backward(ctx, v_means_c, v_covars_c):
        means, covars, viewmats = ctx.saved_tensors
        C, N = viewmats.shape[0], means.shape[0]
        need = ctx.needs_input_grad
        v_means = torch.empty_like(means) if need[0] else None
        v_covars = torch.empty_like(covars) if need[1] else None
        v_viewmats = torch.zeros_like(viewmats) if need[2] else None
        with _device_of(means):
            B.call("gs_world_to_cam_bwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(viewmats),
                   B.ptr(_f32c(v_means_c)) if v_means_c is not None else None,
                   B.ptr(_f32c(v_covars_c)) if v_covars_c is not None else None,
                   B.ptr(v_means), B.ptr(v_covars), B.ptr(v_viewmats), _stream(means))
        return v_means, v_covars, v_viewmats


def proj(
    means: Tensor,  # [C, N, 3]
    covars: Tensor,  # [C, N, 3, 3]
    Ks: Tensor,  # [C, 3, 3]
    width: int,
    height: int,
    camera_model: Literal["pinhole", "ortho", "fisheye"] = "pinhole",
) -> Tuple[Tensor, Tensor]:
    """Projection of Gaussians (perspective, orthographic or fisheye).

    Returns (means2d [C, N, 2], covars2d [C, N, 2, 2])."""
    C, N, _ = means.shape
    assert means.shape == (C, N, 3), means.size()
    assert covars.shape == (C, N, 3, 3), covars.size()
    assert Ks.shape == (C, 3, 3), Ks.size()
    assert camera_model in _CAMERA_MODELS, camera_model
    return _Proj.apply(means.contiguous(), covars.contiguous(), Ks.contiguous(), width, height, camera_model)


def persp_proj(means: Tensor, covars: Tensor, Ks: Tensor, width: int, height: int) -> Tuple[Tensor, Tensor]:
    """DEPRECATED (as in the reference, _wrapper.py:118-139): use ``proj`` with camera_model="pinhole"."""
    import warnings

    warnings.warn("persp_proj is deprecated and will be removed in a future release. Use proj with ortho=False instead.",
                  DeprecationWarning)
    return proj(means, covars, Ks, width, height, "pinhole")


class _Proj(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, covars, Ks, width, height, camera_model):
        _require_gpu(means, "proj")
        means, covars, Ks = _f32c(means), _f32c(covars), _f32c(Ks)
        C, N = means.shape[0], means.shape[1]
        means2d = torch.empty((C, N, 2), device=means.device)
        covars2d = torch.empty((C, N, 2, 2), device=means.device)
        with _device_of(means):
            B.call("gs_proj_fwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(Ks), int(width), int(height),
                   _CAMERA_MODELS[camera_model], B.ptr(means2d), B.ptr(covars2d), _stream(means))
        ctx.save_for_backward(means, covars, Ks)
        ctx.cfg = (int(width), int(height), _CAMERA_MODELS[camera_model])
        return means2d, covars2d

    @staticmethod
    def backward(ctx, v_means2d, v_covars2d):
        means, covars, Ks = ctx.saved_tensors
        width, height, model = ctx.cfg
        C, N = means.shape[0], means.shape[1]
        v_means2d = _f32c(v_means2d) if v_means2d is not None else torch.zeros((C, N, 2), device=means.device)
        v_covars2d = _f32c(v_covars2d) if v_covars2d is not None else torch.zeros((C, N, 2, 2), device=means.device)
        v_means = torch.empty_like(means)
        v_covars = torch.empty_like(covars)
        with _device_of(means):
            B.call("gs_proj_bwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(Ks), width, height, model, B.ptr(v_means2d),
                   B.ptr(v_covars2d), B.ptr(v_means), B.ptr(v_covars), _stream(means))
        return v_means, v_covars, None, None, None, None


@torch.no_grad()
def rasterize_to_indices_in_range(
    range_start: int,
    range_end: int,
    transmittances: Tensor,  # [C, image_height, image_width]
    means2d: Tensor,  # [C, N, 2]
    conics: Tensor,  # [C, N, 3]
    opacities: Tensor,  # [C, N]
    image_width: int,
    image_height: int,
    tile_size: int,
    isect_offsets: Tensor,  # [C, tile_height, tile_width]
    flatten_ids: Tensor,  # [n_isects]
) -> Tuple[Tensor, Tensor, Tensor]:
    """Rasterizes the list batches ``[range_start, range_end)`` (one batch = tile_size^2 sorted entries per
    tile) and returns only the indices: (gaussian_ids, pixel_ids, camera_ids), flattened [M] int64."""
    C, N, _ = means2d.shape
    assert conics.shape == (C, N, 3), conics.shape
    assert opacities.shape == (C, N), opacities.shape
    assert isect_offsets.shape[0] == C, isect_offsets.shape
    tile_height, tile_width = isect_offsets.shape[1:3]
    assert tile_height * tile_size >= image_height, f"Assert Failed: {tile_height} * {tile_size} >= {image_height}"
    assert tile_width * tile_size >= image_width, f"Assert Failed: {tile_width} * {tile_size} >= {image_width}"
    _require_gpu(means2d, "rasterize_to_indices_in_range")
    dev = means2d.device
    n_isects = int(flatten_ids.shape[0])
    empty = torch.empty((0,), dtype=torch.int64, device=dev)
    if n_isects == 0:
        return empty, empty.clone(), empty.clone()
    # saturate like the reference's uint32 arguments (callers pass e.g. 1e10 for "everything")
    rs = int(min(max(range_start, 0), 0xFFFFFFFF))
    re = int(min(max(range_end, 0), 0xFFFFFFFF))
    trans = _f32c(transmittances)
    means2d, conics, opacities = _f32c(means2d), _f32c(conics), _f32c(opacities)
    offsets = isect_offsets.contiguous().to(torch.int32)
    flat = flatten_ids.contiguous().to(torch.int32)
    cnts = torch.zeros((C * image_height * image_width,), dtype=torch.int32, device=dev)
    common = (rs, re, C, N, n_isects, B.ptr(means2d), B.ptr(conics), B.ptr(opacities), int(image_width), int(image_height),
              int(tile_size), int(tile_width), int(tile_height), B.ptr(offsets), B.ptr(flat), B.ptr(trans))
    with _device_of(means2d):
        B.call("gs_rasterize_indices_count", *common, B.ptr(cnts), _stream(means2d))
        cumsum = torch.cumsum(cnts, 0, dtype=torch.int32)
        n_elems = int(cumsum[-1].item())
        starts = (cumsum - cnts).contiguous()
        gaussian_ids = torch.empty((n_elems,), dtype=torch.int64, device=dev)
        pixel_ids = torch.empty((n_elems,), dtype=torch.int64, device=dev)
        if n_elems:
            B.call("gs_rasterize_indices_fill", *common, B.ptr(starts), B.ptr(gaussian_ids), B.ptr(pixel_ids), _stream(means2d))
    out_pixel_ids = pixel_ids % (image_width * image_height)
    out_camera_ids = pixel_ids // (image_width * image_height)
    return gaussian_ids, out_pixel_ids, out_camera_ids


def accumulate(
    means2d: Tensor,  # [C, N, 2]
    conics: Tensor,  # [C, N, 3]
    opacities: Tensor,  # [C, N]
    colors: Tensor,  # [C, N, channels]
    gaussian_ids: Tensor,  # [M]
    pixel_ids: Tensor,  # [M]
    camera_ids: Tensor,  # [M]
    image_width: int,
    image_height: int,
) -> Tuple[Tensor, Tensor]:
    """Alpha compositing over an explicit list of (gaussian, pixel, camera) intersections -- the reference's ``gsplat.accumulate``
    (gsplat/cuda/_torch_impl.py:432-519, torch ops + nerfacc there; here ``gs_accumulate_fwd`` / ``_bwd``, csrc/unfused.hip).  The
    intersections come from ``rasterize_to_indices_in_range`` (grouped by ray, front to back).  Differentiable in means2d, conics,
    opacities and colors.  Returns ``(renders [C, image_height, image_width, channels], alphas [C, image_height, image_width, 1])``."""
    C, N = means2d.shape[:2]
    assert means2d.shape == (C, N, 2) and conics.shape == (C, N, 3) and opacities.shape == (C, N), (means2d.shape, conics.shape, opacities.shape)
    assert colors.dim() == 3 and colors.shape[:2] == (C, N), colors.shape
    M = gaussian_ids.shape[0]
    assert gaussian_ids.shape == pixel_ids.shape == camera_ids.shape == (M,), (gaussian_ids.shape, pixel_ids.shape, camera_ids.shape)
    return _Accumulate.apply(means2d, conics, opacities, colors, gaussian_ids, pixel_ids, camera_ids, int(image_width), int(image_height))


class _Accumulate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, opacities, colors, gaussian_ids, pixel_ids, camera_ids, width, height):
        _require_gpu(means2d, "accumulate")
        means2d, conics, opacities, colors = _f32c(means2d), _f32c(conics), _f32c(opacities), _f32c(colors)
        gids, pids, cids = (t.contiguous().to(torch.int64) for t in (gaussian_ids, pixel_ids, camera_ids))
        C, N, channels = colors.shape
        M = gids.shape[0]
        dev = means2d.device
        renders = torch.zeros((C, height, width, channels), dtype=torch.float32, device=dev)
        alphas = torch.zeros((C, height, width, 1), dtype=torch.float32, device=dev)
        alpha_buf = torch.empty((M,), dtype=torch.float32, device=dev)
        weights = torch.empty((M,), dtype=torch.float32, device=dev)
        with _device_of(means2d):
            B.call("gs_accumulate_fwd", M, C, N, channels, B.ptr(means2d), B.ptr(conics), B.ptr(opacities), B.ptr(colors), B.ptr(gids),
                   B.ptr(pids), B.ptr(cids), width, height, B.ptr(alpha_buf), B.ptr(weights), B.ptr(renders), B.ptr(alphas), _stream(means2d))
        ctx.save_for_backward(means2d, conics, opacities, colors, gids, pids, cids, alpha_buf, weights)
        ctx.size = (width, height)
        ctx.set_materialize_grads(False)
        return renders, alphas

    @staticmethod
    def backward(ctx, v_renders, v_alphas):
        means2d, conics, opacities, colors, gids, pids, cids, alpha_buf, weights = ctx.saved_tensors
        C, N, channels = colors.shape
        M = gids.shape[0]
        need = ctx.needs_input_grad
        v_renders = _f32c(v_renders) if v_renders is not None else None
        v_alphas = _f32c(v_alphas) if v_alphas is not None else None
        outs = [torch.zeros_like(t) if need[i] else None for i, t in enumerate((means2d, conics, opacities, colors))]
        if M and (v_renders is not None or v_alphas is not None):
            scratch = torch.empty((M,), dtype=torch.float32, device=means2d.device)
            with _device_of(means2d):
                B.call("gs_accumulate_bwd", M, C, N, channels, B.ptr(means2d), B.ptr(conics), B.ptr(opacities), B.ptr(colors), B.ptr(gids),
                       B.ptr(pids), B.ptr(cids), ctx.size[0], ctx.size[1], B.ptr(alpha_buf), B.ptr(weights), B.ptr(v_renders), B.ptr(v_alphas),
                       B.ptr(scratch), *[B.ptr(o) for o in outs], _stream(means2d))
        return (*outs, None, None, None, None, None)


# ---------------------------------------------------------------------------
# quat/scale -> covar/preci  (reference _wrapper.py:76-115, 646-706)
# ---------------------------------------------------------------------------
def quat_scale_to_covar_preci(
    quats: Tensor,  # [N, 4],
    scales: Tensor,  # [N, 3],
    compute_covar: bool = True,
    compute_preci: bool = True,
    triu: bool = False,
) -> Tuple[Optional[Tensor], Optional[Tensor]]:
    """Converts quaternions and scales to covariance and precision matrices.

    Returns (covars, precis): [N,3,3] each, or [N,6] upper-triangular when ``triu``.
    """
    assert quats.dim() == 2 and quats.size(1) == 4, quats.size()
    assert scales.dim() == 2 and scales.size(1) == 3, scales.size()
    quats = quats.contiguous()
    scales = scales.contiguous()
    covars, precis = _QuatScaleToCovarPreci.apply(quats, scales, compute_covar, compute_preci, triu)
    return covars if compute_covar else None, precis if compute_preci else None


class _QuatScaleToCovarPreci(torch.autograd.Function):
    @staticmethod
    def forward(ctx, quats, scales, compute_covar, compute_preci, triu):
        _require_gpu(quats, "quat_scale_to_covar_preci")
        quats, scales = _f32c(quats), _f32c(scales)
        N = quats.shape[0]
        shape = (N, 6) if triu else (N, 3, 3)
        covars = torch.empty(shape, device=quats.device) if compute_covar else None
        precis = torch.empty(shape, device=quats.device) if compute_preci else None
        with _device_of(quats):
            B.call("gs_quat_scale_to_covar_preci_fwd", N, B.ptr(quats), B.ptr(scales), int(triu), B.ptr(covars),
                   B.ptr(precis), _stream(quats))
        ctx.save_for_backward(quats, scales)
        ctx.flags = (compute_covar, compute_preci, triu)
        return covars, precis

    @staticmethod
    def backward(ctx, v_covars, v_precis):
        quats, scales = ctx.saved_tensors
        compute_covar, compute_preci, triu = ctx.flags
        v_covars = _f32c(v_covars) if (compute_covar and v_covars is not None) else None
        v_precis = _f32c(v_precis) if (compute_preci and v_precis is not None) else None
        N = quats.shape[0]
        v_quats = torch.empty_like(quats)
        v_scales = torch.empty_like(scales)
        with _device_of(quats):
            B.call("gs_quat_scale_to_covar_preci_bwd", N, B.ptr(quats), B.ptr(scales), int(triu), B.ptr(v_covars),
                   B.ptr(v_precis), B.ptr(v_quats), B.ptr(v_scales), _stream(quats))
        return v_quats, v_scales, None, None, None


# ---------------------------------------------------------------------------
# fully fused projection  (reference _wrapper.py:203-339, 775-898, 1031-1223)
# ---------------------------------------------------------------------------
def fully_fused_projection(
    means: Tensor,  # [N, 3]
    covars: Optional[Tensor],  # [N, 6] or None
    quats: Optional[Tensor],  # [N, 4] or None
    scales: Optional[Tensor],  # [N, 3] or None
    viewmats: Tensor,  # [C, 4, 4]
    Ks: Tensor,  # [C, 3, 3]
    width: int,
    height: int,
    eps2d: float = 0.3,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    packed: bool = False,
    sparse_grad: bool = False,
    calc_compensations: bool = False,
    camera_model: Literal["pinhole", "ortho", "fisheye"] = "pinhole",
    _means_alias: bool = False,  # (internal, unpacked only) also return the means, routed through the projection's backward
):
    """Projects Gaussians to 2D.

    Unpacked: returns (radii [C,N] i32, means2d [C,N,2], depths [C,N], conics [C,N,3],
    compensations [C,N] or None); only entries with radii > 0 are valid.
    Packed: returns (camera_ids [nnz] i64, gaussian_ids [nnz] i64, radii [nnz], means2d,
    depths, conics, compensations) with all entries valid.
    """
    C = viewmats.size(0)
    N = means.size(0)
    assert means.size() == (N, 3), means.size()
    assert viewmats.size() == (C, 4, 4), viewmats.size()
    assert Ks.size() == (C, 3, 3), Ks.size()
    means = means.contiguous()
    if covars is not None:
        assert covars.size() == (N, 6), covars.size()
        covars = covars.contiguous()
    else:
        assert quats is not None, "covars or quats is required"
        assert scales is not None, "covars or scales is required"
        assert quats.size() == (N, 4), quats.size()
        assert scales.size() == (N, 3), scales.size()
        quats = quats.contiguous()
        scales = scales.contiguous()
    if sparse_grad:
        assert packed, "sparse_grad is only supported when packed is True"
    assert camera_model in _CAMERA_MODELS, camera_model

    viewmats = viewmats.contiguous()
    Ks = Ks.contiguous()
    if packed:
        return _FullyFusedProjectionPacked.apply(
            means, covars, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
            radius_clip, sparse_grad, calc_compensations, camera_model,
        )
    return _FullyFusedProjection.apply(
        means, covars, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane,
        radius_clip, calc_compensations, camera_model, _means_alias,
    )


class _FullyFusedProjection(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, covars, quats, scales, viewmats, Ks, width, height, eps2d, near_plane,
                far_plane, radius_clip, calc_compensations, camera_model="pinhole", means_alias=False):
        _require_gpu(means, "fully_fused_projection")
        means_in = means
        means, covars, quats, scales = _f32c(means), _f32c(covars), _f32c(quats), _f32c(scales)
        viewmats, Ks = _f32c(viewmats), _f32c(Ks)
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        radii = torch.empty((C, N), dtype=torch.int32, device=dev)
        means2d = torch.empty((C, N, 2), dtype=torch.float32, device=dev)
        depths = torch.empty((C, N), dtype=torch.float32, device=dev)
        conics = torch.empty((C, N, 3), dtype=torch.float32, device=dev)
        # the reference zero-initialises compensations only (fwd.cu:242-245)
        compensations = torch.zeros((C, N), dtype=torch.float32, device=dev) if calc_compensations else None
        cm = _CAMERA_MODELS[camera_model]
        with _device_of(means):
            B.call("gs_projection_fwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(quats), B.ptr(scales),
                   B.ptr(viewmats), B.ptr(Ks), int(width), int(height), float(eps2d), float(near_plane),
                   float(far_plane), float(radius_clip), cm, B.ptr(radii), B.ptr(means2d), B.ptr(depths),
                   B.ptr(conics), B.ptr(compensations), _stream(means))
        ctx.save_for_backward(means, covars, quats, scales, viewmats, Ks, radii, conics, compensations)
        ctx.width, ctx.height, ctx.eps2d, ctx.cm = width, height, eps2d, cm
        ctx.mark_non_differentiable(radii)
        ctx.set_materialize_grads(False)  # unused outputs (radii, depths in RGB mode) arrive as None, not as zero tensors
        if means_alias:
            # the means handed through: what a second consumer (the SH view directions) sends back for them arrives in
            # THIS node's backward, where the projection kernel adds it while writing v_means -- instead of autograd
            # summing two [N,3] gradients in a pass of its own
            return radii, means2d, depths, conics, compensations, means_in
        return radii, means2d, depths, conics, compensations

    @staticmethod
    def backward(ctx, v_radii, v_means2d, v_depths, v_conics, v_compensations, v_means_add=None):
        means, covars, quats, scales, viewmats, Ks, radii, conics, compensations = ctx.saved_tensors
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        if v_means2d is None:
            v_means2d = torch.zeros((C, N, 2), dtype=torch.float32, device=dev)
        if v_conics is None:
            v_conics = torch.zeros((C, N, 3), dtype=torch.float32, device=dev)
        (v_means2d, s_m2), (v_conics, s_cn) = _row_strided(v_means2d, 2), _row_strided(v_conics, 3)
        v_depths = _f32c(v_depths) if v_depths is not None else None
        v_compensations = _f32c(v_compensations) if v_compensations is not None else None
        need = ctx.needs_input_grad
        # rows are fully written by the kernel -> empty, not zeros
        v_means = torch.empty_like(means) if need[0] else None
        v_covars = torch.empty_like(covars) if (covars is not None and need[1]) else None
        v_quats = torch.empty_like(quats) if (quats is not None and need[2]) else None
        v_scales = torch.empty_like(scales) if (scales is not None and need[3]) else None
        v_viewmats = torch.zeros_like(viewmats) if need[4] else None
        with _device_of(means):
            B.call("gs_projection_bwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(quats), B.ptr(scales),
                   B.ptr(viewmats), B.ptr(Ks), int(ctx.width), int(ctx.height), float(ctx.eps2d), ctx.cm,
                   B.ptr(radii), B.ptr(conics), B.ptr(compensations), B.ptr(v_means2d), B.ptr(v_depths),
                   B.ptr(v_conics), B.ptr(v_compensations), B.ptr(v_means), B.ptr(v_covars), B.ptr(v_quats),
                   B.ptr(v_scales), B.ptr(v_viewmats), s_m2, s_cn,
                   B.ptr(_f32c(v_means_add)) if (v_means_add is not None and v_means is not None) else None, _stream(means))
        return (v_means, v_covars, v_quats, v_scales, v_viewmats) + (None,) * 10


_FUSE_SH_BWD = os.environ.get("GS_FUSE_SH_BWD", "1") == "1"  # (A/B switch: 0 = gs_sh_view_bwd + gs_projection_rows_bwd as two launches)


# GS_GRAD_PREFILL=0: the per-gaussian gradients are allocated by the projection backward itself instead of behind the
# compositing gradient rows (they then do not keep the C * N * 64-byte row buffer alive while they are held as .grad, and a
# forward under grad mode that never runs backward does not zero-fill ~236 B per gaussian for nothing; ~3 % slower per step)
PREFILL_ENABLED = os.environ.get("GS_GRAD_PREFILL", "1") != "0"


_FAST_MAX_CHANNELS = 32  # channel counts the tile forward / segmented backward cover (csrc/rasterize.hip: FAST_MAX_CHANNELS)


# The order in which the per-gaussian gradients of ONE rasterization() call are carved out of their common buffer (GradPrefill.carve).
# distributed.all_reduce_splat_grads reduces that buffer in place when a rank's gradients lie in exactly this order
# (distributed._CARVE_RANK is derived from it), so both autograd nodes that build a request go through prefill_request().
PREFILL_ORDER = ("means", "covars", "quats", "scales", "opacities", "colors", "sh", "sh_rest", "motion", "omega", "trbf_center", "trbf_scale")


def prefill_request(items) -> list:
    """[(key, shape)] of the wanted tensors among ``items`` = [(key, tensor or None, wanted)], in PREFILL_ORDER."""
    req = [(key, tuple(t.shape)) for key, t, flag in items if t is not None and flag]
    return sorted(req, key=lambda kv: PREFILL_ORDER.index(kv[0]))


class GradPrefill:
    """Hand-over between the two autograd nodes of ONE ``rasterization()`` call (not in the reference).

    The per-gaussian gradients the projection backward returns (``v_sh`` 192 B per gaussian at degree 3, means / quats /
    scales / opacities 44 B) are dense tensors whose rows are exact zeros for every gaussian no camera sees -- 71 % of
    them at BASELINE config 2, 137 of the 193 MB the SH backward writes.  With a ``GradPrefill`` the projection node lists
    in its FORWARD what its backward will return (``request``), the compositing forward allocates those tensors behind
    its own gradient rows in one buffer and zero-fills the lot as the side job of its tile workgroups (``zero_fill`` of
    ``gs_rasterize_fwd``: the memory pipes idle while the chip composites), and the projection backward then stores
    only the rows of visible gaussians (``outputs_prefilled`` of ``gs_sh_view_bwd`` / ``gs_projection_rows_bwd``).
    ``parts`` is consumed by the first backward; a repeated one (retain_graph) allocates and writes everything itself."""

    __slots__ = ("request", "parts")

    def __init__(self):
        self.request = []  # [(key, shape)] filled by _ProjectRows.forward
        self.parts = {}  # key -> zero-filled tensor, made by _RasterizeToPixels.forward

    def floats(self) -> int:
        return sum(_pad64(math.prod(shape)) for _, shape in self.request)

    def carve(self, buf: Tensor, offset: int) -> None:
        """Cut the requested tensors out of the flat fp32 ``buf`` starting at ``offset`` (256-byte aligned pieces)."""
        self.parts = {}
        for key, shape in self.request:
            n = math.prod(shape)
            self.parts[key] = buf[offset:offset + n].view(shape)
            offset += _pad64(n)

    def take(self) -> dict:
        parts, self.parts = self.parts, {}
        return parts


def _pad64(n: int) -> int:
    return (n + 63) // 64 * 64


def project_rows(
    means: Tensor,  # [N, 3]
    covars: Optional[Tensor],  # [N, 6] or None
    quats: Optional[Tensor],  # [N, 4] or None
    scales: Optional[Tensor],  # [N, 3] or None
    viewmats: Tensor,  # [C, 4, 4]
    Ks: Tensor,  # [C, 3, 3]
    width: int,
    height: int,
    opacities: Tensor,  # [N]
    colors: Optional[Tensor] = None,  # [N, 3] post-activation colours shared by all cameras, or None
    eps2d: float = 0.3,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    antialiased: bool = False,
    camera_model: Literal["pinhole", "ortho", "fisheye"] = "pinhole",
    sh_coeffs: Optional[Tensor] = None,  # [N, K, 3] SH coefficients shared by all cameras (instead of ``colors``)
    sh_degree: Optional[int] = None,
    sh_rest: Optional[Tensor] = None,  # SPLIT coefficients: ``sh_coeffs`` is the DC band [N, 1, 3], this [N, K-1, 3]
    prefill: Optional[GradPrefill] = None,  # see GradPrefill; hand the same object to ``rasterize_to_pixels``
    sh_mask=None,  # (mask_logits with N elements, temperature, binary): the shN mask applied on the fly (split rows only)
    dynamic=None,  # dynamic.DynamicSlice: the temporal slice (+ opt-in activations / round quantizer) evaluated by the projection itself
):
    """``fully_fused_projection`` in ROW form, what ``rasterization`` uses for unpacked batches: the same projection, but
    every (camera, gaussian) pair gets one 64-byte splat row (include/gsplat_hip.h) that the compositing kernels fetch
    whole.  Folded in: the per-view opacities (``opacities.repeat(C, 1)``, times the antialias compensation) and,
    when given, the per-view colours (``colors.expand(C, -1, -1)``) of reference rendering.py:327-335, 386 -- or, from
    ``sh_coeffs``, the SH colours ``clamp_min(spherical_harmonics(sh_degree, means - campos, sh_coeffs, radii > 0) + 0.5, 0)``
    of rendering.py:368-392, evaluated in the same pass (bit-identical to ``spherical_harmonics_view``).

    Returns ``(radii [C,N] i32, means2d [C,N,2], depths [C,N], conics [C,N,3], opacities [C,N], colors [C,N,3] | None,
    rows [C,N,16])``: means2d / conics / opacities / colors are COLUMN VIEWS of ``rows`` (only defined where radii > 0,
    like the reference's means2d / conics), differentiable like the separate tensors of ``fully_fused_projection``."""
    C, N = viewmats.size(0), means.size(0)
    assert means.size() == (N, 3), means.size()
    assert viewmats.size() == (C, 4, 4), viewmats.size()
    assert Ks.size() == (C, 3, 3), Ks.size()
    assert opacities.size() == (N,), opacities.size()
    if covars is not None:
        assert covars.size() == (N, 6), covars.size()
        covars = covars.contiguous()
    else:
        assert quats is not None and scales is not None, "covars or (quats, scales) is required"
        assert quats.size() == (N, 4) and scales.size() == (N, 3), (quats.size(), scales.size())
        quats, scales = quats.contiguous(), scales.contiguous()
    if colors is not None:
        assert colors.size() == (N, 3), colors.size()
        colors = colors.contiguous()
    if sh_coeffs is not None:
        assert colors is None and sh_degree is not None, "sh_coeffs come with sh_degree and without colors"
        assert sh_coeffs.dim() == 3 and sh_coeffs.shape[0] == N and sh_coeffs.shape[2] == 3, sh_coeffs.shape
        K = sh_coeffs.shape[1]
        if sh_rest is not None:
            assert sh_coeffs.shape[1] == 1 and sh_rest.dim() == 3 and sh_rest.shape[0] == N and sh_rest.shape[2] == 3 \
                and sh_rest.shape[1] >= 1, (sh_coeffs.shape, sh_rest.shape)
            K += sh_rest.shape[1]
            sh_rest = sh_rest.contiguous()
        assert (sh_degree + 1) ** 2 <= K, (sh_degree, K)
        sh_coeffs = sh_coeffs.contiguous()
    else:
        assert sh_rest is None
    assert camera_model in _CAMERA_MODELS, camera_model
    mask_logits, mask_cfg = None, None
    if sh_mask is not None:
        assert sh_rest is not None, "the shN mask needs split coefficients (sh_rest)"
        mask_logits, mask_cfg = sh_mask[0], (float(sh_mask[1]), bool(sh_mask[2]))
        assert mask_logits.numel() == N, (mask_logits.shape, N)
    if dynamic is not None:
        assert covars is None and sh_coeffs is None, "dynamic splats: quats + scales and [N, 3] colours (or none) only"
        assert not viewmats.requires_grad, "project_rows(dynamic=...): no camera-pose gradients on the fused route (rasterization() falls back)"
        dynamic.check(N)
        return _ProjectRows.apply(means.contiguous(), covars, quats, scales, viewmats.contiguous(), Ks.contiguous(),
                                  opacities.contiguous(), colors, sh_coeffs, sh_rest, mask_logits, width, height, eps2d, near_plane, far_plane,
                                  radius_clip, antialiased, camera_model, sh_degree, prefill, mask_cfg, dynamic.motion, dynamic.omega,
                                  dynamic.trbf_center, dynamic.trbf_scale, dynamic)
    return _ProjectRows.apply(means.contiguous(), covars, quats, scales, viewmats.contiguous(), Ks.contiguous(),
                              opacities.contiguous(), colors, sh_coeffs, sh_rest, mask_logits, width, height, eps2d, near_plane, far_plane,
                              radius_clip, antialiased, camera_model, sh_degree, prefill, mask_cfg)


def _grad_rows_of(parts, shape, device):
    """The [C,N,16] gradient-row buffer behind the gradients autograd hands to ``_ProjectRows.backward``.  ``parts`` =
    [(gradient or None, first column, width)].  When they are the column views ``_RasterizeToPixels.backward`` returns (one
    buffer, splat-row columns) that buffer is used IN PLACE; anything else (a loss on meta["means2d"] itself, gradients
    autograd had to add up, missing ones) is assembled into a fresh zero-filled buffer."""
    base = None
    ok = True
    lead = tuple(shape)
    n_rows = 1
    for d in lead:
        n_rows *= d

    def is_row_view(g, width):
        if g.dtype != torch.float32:
            return False
        if g.dim() == len(lead) + 1:
            if tuple(g.shape) != lead + (width,) or g.stride(-1) != 1:
                return False
        elif g.dim() != len(lead) or width != 1 or tuple(g.shape) != lead:
            return False
        expect = ROW
        for d in range(len(lead) - 1, -1, -1):  # the leading dims collapse to one row index
            if lead[d] != 1 and g.stride(d) != expect:
                return False
            expect *= lead[d]
        return True

    for g, col, width in parts:
        if g is None:
            ok = False
            continue
        p0 = g.data_ptr() - 4 * col
        if n_rows > 0 and is_row_view(g, width) and (base is None or base == p0) and p0 % 16 == 0:
            base = p0
        else:
            ok = False
    if ok and base is not None:
        return base, None
    G = torch.zeros(tuple(shape) + (ROW,), dtype=torch.float32, device=device)
    for g, col, width in parts:
        if g is not None:
            G[..., col:col + width] = g.reshape(tuple(shape) + (width,))
    return G.data_ptr(), G


def dyn_prefill_items(dyn_ctx, need, first: int):
    """(key, tensor, wanted) of the four extra inputs of the dynamic route (motion, omega, trbf_center, trbf_scale) for GradPrefill."""
    _, dt = dyn_ctx
    return (("motion", dt[0], need[first]), ("omega", dt[1], need[first + 1]), ("trbf_center", dt[2], need[first + 2]),
            ("trbf_scale", dt[3], need[first + 3]))


def _project_rows_dyn_bwd(ctx, dyn_ctx, need, out, prefilled, g_ptr, g_keep, v_depths):
    """``_ProjectRows.backward`` of the dynamic route: gs_projection_rows_dyn_bwd = the projection VJP + the slice / activation /
    STE VJPs for the gaussians some camera saw; returns the gradient tuple in the order of ``_ProjectRows.forward``'s inputs."""
    means, covars, quats, scales, viewmats, Ks, opacities, radii, rows, sh_coeffs, sh_rest = ctx.saved_tensors
    dyn, dt = dyn_ctx
    motion, omega, center, tscale = dt
    C, N = viewmats.shape[0], means.shape[0]
    f = ctx.dyn_first
    if need[4]:
        raise RuntimeError("rasterization(dynamic=...): camera-pose gradients are not available on the fused dynamic route")
    v_depths = _f32c(v_depths) if v_depths is not None else None
    v_means = out("means", means) if need[0] else None
    v_quats = out("quats", quats) if need[2] else None
    v_scales = out("scales", scales) if need[3] else None
    v_opac = out("opacities", opacities) if need[6] else None
    v_colors = (out("colors", torch.empty(0)) if prefilled else torch.empty((N, 3), dtype=torch.float32, device=means.device)) \
        if (ctx.has_colors and need[7]) else None
    v_motion = out("motion", motion) if need[f] else None
    v_omega = out("omega", omega) if need[f + 1] else None
    v_center = out("trbf_center", center) if need[f + 2] else None
    v_tscale = out("trbf_scale", tscale) if need[f + 3] else None
    with _device_of(means):
        B.call("gs_projection_rows_dyn_bwd", C, N, B.ptr(means), B.ptr(quats), B.ptr(scales), *dyn.c_args(dt), B.ptr(viewmats), B.ptr(Ks),
               int(ctx.width), int(ctx.height), float(ctx.eps2d), ctx.cm, B.ptr(radii), B.ptr(rows), g_ptr, B.ptr(v_depths), B.ptr(opacities),
               int(ctx.antialiased), B.ptr(v_means), B.ptr(v_quats), B.ptr(v_scales), B.ptr(v_motion), B.ptr(v_omega), B.ptr(v_center),
               B.ptr(v_tscale), B.ptr(v_opac), B.ptr(v_colors), int(prefilled), _stream(means))
    del g_keep
    return (v_means, None, v_quats, v_scales, None, None, v_opac, v_colors, None, None, None) + (None,) * 11 + (
        v_motion, v_omega, v_center, v_tscale, None)


class _ProjectRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, covars, quats, scales, viewmats, Ks, opacities, colors, sh_coeffs, sh_rest, mask_logits, width, height,
                eps2d, near_plane, far_plane, radius_clip, antialiased, camera_model="pinhole", sh_degree=None, prefill=None,
                mask_cfg=None, dyn_motion=None, dyn_omega=None, dyn_center=None, dyn_tscale=None, dyn=None):
        _require_gpu(means, "project_rows")
        means, covars, quats, scales = _f32c(means), _f32c(covars), _f32c(quats), _f32c(scales)
        viewmats, Ks, opacities, colors = _f32c(viewmats), _f32c(Ks), _f32c(opacities), _f32c(colors)
        sh_coeffs, sh_rest, mask_logits = _f32c(sh_coeffs), _f32c(sh_rest), _f32c(mask_logits)
        m_temp, m_bin = mask_cfg if mask_logits is not None else (1.0, False)
        sh_K = (sh_coeffs.shape[1] + (sh_rest.shape[1] if sh_rest is not None else 0)) if sh_coeffs is not None else 0
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        radii = torch.empty((C, N), dtype=torch.int32, device=dev)
        depths = torch.empty((C, N), dtype=torch.float32, device=dev)
        rows = torch.empty((C, N, ROW), dtype=torch.float32, device=dev)  # (torch's allocator aligns to 512 bytes)
        cm = _CAMERA_MODELS[camera_model]
        ctx.dyn = None
        if dyn is not None:
            # dynamic splats: quantizer -> activation -> temporal slice in the projection's load phase (csrc/projection_dyn.hip)
            dtens = dyn.bind(quats, scales, opacities, colors, dyn_motion, dyn_omega, dyn_center, dyn_tscale)
            with _device_of(means):
                B.call("gs_projection_rows_dyn_fwd", C, N, B.ptr(means), B.ptr(quats), B.ptr(scales), *dyn.c_args(dtens, fwd_on=dev, N=N), B.ptr(viewmats),
                       B.ptr(Ks), int(width), int(height), float(eps2d), float(near_plane), float(far_plane), float(radius_clip), cm,
                       B.ptr(opacities), B.ptr(colors), int(bool(antialiased)), 0, 0, 0, None, None, B.ptr(radii), B.ptr(depths),
                       B.ptr(rows), _stream(means))
            ctx.dyn = (dyn, dtens)
        else:
          with _device_of(means):
            B.call("gs_projection_rows_fwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(quats), B.ptr(scales),
                   B.ptr(viewmats), B.ptr(Ks), int(width), int(height), float(eps2d), float(near_plane),
                   float(far_plane), float(radius_clip), cm, B.ptr(opacities), B.ptr(colors), int(bool(antialiased)),
                   B.ptr(sh_coeffs), B.ptr(sh_rest), sh_K, int(sh_degree or 0), B.ptr(mask_logits), float(m_temp), int(m_bin),
                   0, 0, 0, None, None, B.ptr(radii), B.ptr(depths), B.ptr(rows), _stream(means))
        ctx.save_for_backward(means, covars, quats, scales, viewmats, Ks, opacities, radii, rows, sh_coeffs, sh_rest)
        ctx.mask = (mask_logits, float(m_temp), bool(m_bin)) if mask_logits is not None else None
        ctx.width, ctx.height, ctx.eps2d, ctx.cm, ctx.antialiased = width, height, eps2d, cm, bool(antialiased)
        ctx.has_colors, ctx.sh_degree = colors is not None, (int(sh_degree) if sh_coeffs is not None else None)
        ctx.prefill = None
        ctx.dyn_first = 22  # position of dyn_motion among this Function's inputs (_StepProject overrides it)
        need = ctx.needs_input_grad
        if prefill is not None and (any(need[:10]) or (dyn is not None and any(need[22:26]))) and not need[4] and N > 0:
            # what the backward will return per gaussian, for the compositing forward to allocate and zero-fill
            req = prefill_request((("means", means, need[0]), ("covars", covars, need[1]), ("quats", quats, need[2]),
                                   ("scales", scales, need[3]), ("opacities", opacities, need[6]), ("colors", colors, need[7]),
                                   ("sh", sh_coeffs, need[8]), ("sh_rest", sh_rest, need[9]))
                                  + (dyn_prefill_items(ctx.dyn, need, 22) if ctx.dyn is not None else ()))
            prefill.request = req
            ctx.prefill = prefill
        ctx.mark_non_differentiable(radii, rows)
        ctx.set_materialize_grads(False)  # unused outputs (depths in RGB mode, ...) arrive as None, not as zero tensors
        has_col = colors is not None or sh_coeffs is not None
        return (radii, rows[..., ROW_MEAN2D:ROW_MEAN2D + 2], depths, rows[..., ROW_CONIC:ROW_CONIC + 3], rows[..., ROW_OPACITY],
                rows[..., ROW_COLOR:ROW_COLOR + 3] if has_col else None, rows)

    @staticmethod
    def backward(ctx, v_radii, v_means2d, v_depths, v_conics, v_opac_cn, v_colors_cn, v_rows):
        means, covars, quats, scales, viewmats, Ks, opacities, radii, rows, sh_coeffs, sh_rest = ctx.saved_tensors
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        parts = [(v_means2d, ROW_MEAN2D, 2), (v_conics, ROW_CONIC, 3), (v_opac_cn, ROW_OPACITY, 1)]
        if ctx.has_colors or sh_coeffs is not None:
            parts.append((v_colors_cn, ROW_COLOR, 3))
        g_ptr, g_keep = _grad_rows_of(parts, (C, N), dev)
        need = ctx.needs_input_grad
        # outputs the compositing forward allocated and zero-filled for this node (GradPrefill): all of them or none
        pre = ctx.prefill.take() if ctx.prefill is not None else {}
        want = [k for k, t, f in (("means", means, need[0]), ("covars", covars, need[1]), ("quats", quats, need[2]),
                                  ("scales", scales, need[3]), ("opacities", opacities, need[6]),
                                  ("colors", True if ctx.has_colors else None, need[7]), ("sh", sh_coeffs, need[8]),
                                  ("sh_rest", sh_rest, need[9])) if t is not None and f]
        dyn = getattr(ctx, "dyn", None)
        if dyn is not None:
            want += [k for k, t, f in dyn_prefill_items(dyn, need, ctx.dyn_first) if f]
        prefilled = bool(pre) and all(k in pre for k in want) and (sh_coeffs is None or (need[8] and (sh_rest is None or need[9])))
        if not prefilled:
            pre = {}

        def out(key, like):
            return pre[key] if prefilled else torch.empty_like(like)

        if dyn is not None:
            return _project_rows_dyn_bwd(ctx, dyn, need, out, prefilled, g_ptr, g_keep, v_depths)

        v_sh = v_rest = v_means_add = v_mask = None
        mask = getattr(ctx, "mask", None)
        mask_args = (None, 1.0, 0, None)
        sh_args = (None, None, 0, 0, None, None)
        if sh_coeffs is not None:
            # the colour columns of the gradient rows go back through the SH evaluation (clamp gate from the colours in the
            # rows); its d/d means (view directions) is added to v_means.  Vectorisable rows and fixed poses: inside the
            # projection backward's own pass (one launch, one pass over radii / means / the two row buffers); otherwise by
            # gs_sh_view_bwd first, whose v_means the projection kernel then adds while it writes its own
            K = sh_coeffs.shape[1] + (sh_rest.shape[1] if sh_rest is not None else 0)
            v_sh = out("sh", sh_coeffs)
            v_rest = out("sh_rest", sh_rest) if sh_rest is not None else None
            fused = (_FUSE_SH_BWD and (3 * K) % 4 == 0 and not need[4] and v_sh.data_ptr() % 16 == 0 and (v_rest is None or v_rest.data_ptr() % 16 == 0)
                     and (sh_rest is not None or sh_coeffs.data_ptr() % 16 == 0))
            if mask is not None:
                if not fused:
                    raise RuntimeError("project_rows: the fused shN mask needs the fused SH backward (3 K % 4 == 0, aligned rows, fixed poses)")
                if need[10] and not mask[2]:
                    v_mask = torch.empty_like(mask[0])
                mask_args = (B.ptr(mask[0]), mask[1], int(mask[2]), B.ptr(v_mask))
            if fused:
                sh_args = (B.ptr(sh_coeffs), B.ptr(sh_rest), K, ctx.sh_degree, B.ptr(v_sh), B.ptr(v_rest))
            else:
                v_means_add = torch.empty_like(means) if need[0] else None
                with _device_of(means):
                    B.call("gs_sh_view_bwd", C, N, K, ctx.sh_degree, B.ptr(means), B.ptr(viewmats), 1, B.ptr(sh_coeffs), B.ptr(sh_rest),
                           B.ptr(radii), rows.data_ptr() + 4 * ROW_COLOR, ROW, g_ptr + 4 * ROW_COLOR, ROW, B.ptr(v_sh), B.ptr(v_rest),
                           B.ptr(v_means_add), None, 0, None, int(prefilled), _stream(means))
        v_depths = _f32c(v_depths) if v_depths is not None else None
        # rows are fully written by the kernel -> empty, not zeros (prefilled: only the visible gaussians' rows are)
        v_means = out("means", means) if need[0] else None
        v_covars = out("covars", covars) if (covars is not None and need[1]) else None
        v_quats = out("quats", quats) if (quats is not None and need[2]) else None
        v_scales = out("scales", scales) if (scales is not None and need[3]) else None
        v_viewmats = torch.zeros_like(viewmats) if need[4] else None
        v_opac = out("opacities", opacities) if need[6] else None
        v_colors = (pre["colors"] if prefilled else torch.empty((N, 3), dtype=torch.float32, device=dev)) if (ctx.has_colors and need[7]) else None
        with _device_of(means):
            B.call("gs_projection_rows_bwd", C, N, B.ptr(means), B.ptr(covars), B.ptr(quats), B.ptr(scales),
                   B.ptr(viewmats), B.ptr(Ks), int(ctx.width), int(ctx.height), float(ctx.eps2d), ctx.cm,
                   B.ptr(radii), B.ptr(rows), g_ptr, B.ptr(v_depths), B.ptr(opacities), int(ctx.antialiased),
                   B.ptr(v_means), B.ptr(v_covars), B.ptr(v_quats), B.ptr(v_scales), B.ptr(v_viewmats), B.ptr(v_opac),
                   B.ptr(v_colors), B.ptr(v_means_add) if v_means is not None else None, *sh_args, *mask_args, int(prefilled),
                   _stream(means))
        if sh_coeffs is not None:
            if not need[8]:
                v_sh = None
            if not need[9]:
                v_rest = None
        del g_keep
        return (v_means, v_covars, v_quats, v_scales, v_viewmats, None, v_opac, v_colors, v_sh, v_rest, v_mask) + (None,) * 11


class _FullyFusedProjectionPacked(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, covars, quats, scales, viewmats, Ks, width, height, eps2d, near_plane,
                far_plane, radius_clip, sparse_grad, calc_compensations, camera_model="pinhole"):
        _require_gpu(means, "fully_fused_projection(packed)")
        means, covars, quats, scales = _f32c(means), _f32c(covars), _f32c(quats), _f32c(scales)
        viewmats, Ks = _f32c(viewmats), _f32c(Ks)
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        cm = _CAMERA_MODELS[camera_model]
        nblocks = (N + 255) // 256
        st = _stream(means)
        common = (C, N, B.ptr(means), B.ptr(covars), B.ptr(quats), B.ptr(scales), B.ptr(viewmats), B.ptr(Ks),
                  int(width), int(height), float(eps2d), float(near_plane), float(far_plane),
                  float(radius_clip), cm)
        with _device_of(means):
            nnz = 0
            if C * N > 0:
                block_cnts = torch.empty(C * nblocks, dtype=torch.int32, device=dev)
                B.call("gs_projection_packed_count", *common, B.ptr(block_cnts), st)
                block_accum = torch.empty_like(block_cnts)
                sb = B.query("gs_cumsum_scratch_bytes", C * nblocks)
                scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
                B.call("gs_cumsum_i32_i32", C * nblocks, B.ptr(block_cnts), B.ptr(block_accum), B.ptr(scratch),
                       sb, st)
                nnz = int(block_accum[-1].item())  # the one host sync (packed_fwd.cu:335)
            indptr = torch.zeros(C + 1, dtype=torch.int32, device=dev)
            camera_ids = torch.empty(nnz, dtype=torch.int64, device=dev)
            gaussian_ids = torch.empty(nnz, dtype=torch.int64, device=dev)
            radii = torch.empty(nnz, dtype=torch.int32, device=dev)
            means2d = torch.empty((nnz, 2), dtype=torch.float32, device=dev)
            depths = torch.empty(nnz, dtype=torch.float32, device=dev)
            conics = torch.empty((nnz, 3), dtype=torch.float32, device=dev)
            compensations = torch.zeros(nnz, dtype=torch.float32, device=dev) if calc_compensations else None
            if nnz > 0:
                B.call("gs_projection_packed_fill", *common, B.ptr(block_accum), B.ptr(indptr), B.ptr(camera_ids),
                       B.ptr(gaussian_ids), B.ptr(radii), B.ptr(means2d), B.ptr(depths), B.ptr(conics),
                       B.ptr(compensations), st)
        ctx.save_for_backward(camera_ids, gaussian_ids, means, covars, quats, scales, viewmats, Ks, conics,
                              compensations)
        ctx.width, ctx.height, ctx.eps2d, ctx.cm, ctx.sparse_grad = width, height, eps2d, cm, sparse_grad
        ctx.mark_non_differentiable(camera_ids, gaussian_ids, radii)
        return camera_ids, gaussian_ids, radii, means2d, depths, conics, compensations

    @staticmethod
    def backward(ctx, v_camera_ids, v_gaussian_ids, v_radii, v_means2d, v_depths, v_conics, v_compensations):
        (camera_ids, gaussian_ids, means, covars, quats, scales, viewmats, Ks, conics,
         compensations) = ctx.saved_tensors
        C, N, nnz = viewmats.shape[0], means.shape[0], camera_ids.shape[0]
        sparse = ctx.sparse_grad
        need = ctx.needs_input_grad
        v_means2d, v_depths, v_conics = _f32c(v_means2d), _f32c(v_depths), _f32c(v_conics)
        v_compensations = _f32c(v_compensations) if v_compensations is not None else None
        dev = means.device

        def buf(like: Optional[Tensor], flag: bool, width_: int) -> Optional[Tensor]:
            if like is None or not flag:
                return None
            if sparse:
                return torch.empty((nnz, width_), dtype=torch.float32, device=dev)
            return torch.zeros((N, width_), dtype=torch.float32, device=dev)

        v_means = buf(means, need[0], 3)
        v_covars = buf(covars, need[1], 6)
        v_quats = buf(quats, need[2], 4)
        v_scales = buf(scales, need[3], 3)
        v_viewmats = torch.zeros_like(viewmats) if need[4] else None
        with _device_of(means):
            B.call("gs_projection_packed_bwd", C, N, nnz, B.ptr(means), B.ptr(covars), B.ptr(quats), B.ptr(scales),
                   B.ptr(viewmats), B.ptr(Ks), int(ctx.width), int(ctx.height), float(ctx.eps2d), ctx.cm,
                   B.ptr(camera_ids), B.ptr(gaussian_ids), B.ptr(conics), B.ptr(compensations), B.ptr(v_means2d),
                   B.ptr(v_depths), B.ptr(v_conics), B.ptr(v_compensations), int(sparse), B.ptr(v_means),
                   B.ptr(v_covars), B.ptr(v_quats), B.ptr(v_scales), B.ptr(v_viewmats), _stream(means))
        if sparse:
            # reference: torch.sparse_coo_tensor(indices=gaussian_ids[None], values, size, is_coalesced=(C==1))
            def coo(v: Optional[Tensor], like: Tensor) -> Optional[Tensor]:
                if v is None:
                    return None
                return torch.sparse_coo_tensor(indices=gaussian_ids[None], values=v, size=like.size(),
                                               is_coalesced=(C == 1))

            v_means = coo(v_means, means)
            v_covars = coo(v_covars, covars) if covars is not None else None
            v_quats = coo(v_quats, quats) if quats is not None else None
            v_scales = coo(v_scales, scales) if scales is not None else None
        return (v_means, v_covars, v_quats, v_scales, v_viewmats) + (None,) * 10


# ---------------------------------------------------------------------------
# tile intersection  (reference _wrapper.py:342-433)
# ---------------------------------------------------------------------------
@torch.no_grad()
def isect_tiles(
    means2d: Tensor,  # [C, N, 2] or [nnz, 2]
    radii: Tensor,  # [C, N] or [nnz]
    depths: Tensor,  # [C, N] or [nnz]
    tile_size: int,
    tile_width: int,
    tile_height: int,
    sort: bool = True,
    packed: bool = False,
    n_cameras: Optional[int] = None,
    camera_ids: Optional[Tensor] = None,
    gaussian_ids: Optional[Tensor] = None,
) -> Tuple[Tensor, Tensor, Tensor]:
    """Maps projected Gaussians to intersecting tiles.

    Returns (tiles_per_gauss i32, isect_ids i64 [n_isects], flatten_ids i32 [n_isects]);
    an id is ``camera_id << (32 + tile_bits) | tile_id << 32 | float_bits(depth)``.
    """
    return isect_tiles_finish(isect_tiles_start(means2d, radii, depths, tile_size, tile_width, tile_height, sort, packed,
                                                n_cameras, camera_ids, gaussian_ids))


def isect_tiles_start(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True, packed=False, n_cameras=None,
                      camera_ids=None, gaussian_ids=None):
    """``isect_tiles`` up to its host read-back (same arguments); hand the result to ``isect_tiles_finish``."""
    if packed:
        nnz = means2d.size(0)
        assert means2d.shape == (nnz, 2), means2d.size()
        assert radii.shape == (nnz,), radii.size()
        assert depths.shape == (nnz,), depths.size()
        assert camera_ids is not None, "camera_ids is required if packed is True"
        assert gaussian_ids is not None, "gaussian_ids is required if packed is True"
        assert n_cameras is not None, "n_cameras is required if packed is True"
        camera_ids = camera_ids.contiguous()
        gaussian_ids = gaussian_ids.contiguous()
        C = n_cameras
        N = 0
        n_elems = nnz
    else:
        C, N, _ = means2d.shape
        assert means2d.shape == (C, N, 2), means2d.size()
        assert radii.shape == (C, N), radii.size()
        assert depths.shape == (C, N), depths.size()
        camera_ids = None
        n_elems = C * N
    return isect_tiles_begin(means2d, radii, depths, tile_size, tile_width, tile_height, sort, C, N, n_elems, camera_ids)


# pinned host buffers the count kernel writes its per-block sums into: taken in isect_tiles_begin, handed back in
# isect_tiles_finish once read (a buffer is never shared by two calls in flight; one whose finish never runs is simply
# garbage-collected).  Re-used so that the steady state makes no pinned allocation.
_PINNED_FREE: dict = {}
_PINNED_DIRECT_MAX = 2048  # block sums a kernel may store straight into pinned host memory (4-byte PCIe writes)


# the depth pre-sort's route: "on" = the bucketed form where it applies (GS_PRESORT=0: always the LSD radix sort);
# "lds_capacity": keys a local sort may hold in LDS (0 = the library's 4096; tests lower it to drive the global-memory route)
_PRESORT = {"on": os.environ.get("GS_PRESORT", "1") != "0", "lds_capacity": 0}


def presort_split_buffer(dev: torch.device) -> Tensor:
    """The int64 buffer ``gs_presort_split`` works in: the 256 splitters in front, its candidate slots behind them."""
    return torch.empty(_SPLIT_ELEMS[0] or _split_elems(), dtype=torch.int64, device=dev)


_SPLIT_ELEMS = [0]


def _split_elems() -> int:
    _SPLIT_ELEMS[0] = int(B.query("gs_presort_split_elems"))
    return _SPLIT_ELEMS[0]


def block_sum_totals(a) -> Tuple[int, int]:
    """(sum of the even entries, sum of the odd entries) of the pinned int32 block-sum array [(intersections, visible)] -- exact, and
    ~10 us instead of the ~50 of ``a.reshape(-1, 2).sum(0)`` (a strided reduction with a dtype conversion), which sat between the
    host's read-back and the binning launch with the GPU waiting (round 6, tools/host_timeline.py).  The non-negative pairs are read as
    int64 words (little endian: even entry = low half, odd entry = high half) and the halves summed separately."""
    w = a.view("int64")
    return int((w & 0xFFFFFFFF).sum()), int((w >> 32).sum())


def _pinned_take(n: int) -> Tensor:
    """A pinned int32 buffer the count kernel stores its block sums into, PRE-SET to -1: every sum is >= 0, so the host sees
    the kernel's progress in the buffer itself (``_SentinelEvent``) and no event has to be recorded behind the kernel -- a
    recorded event is a barrier packet in the queue, ~6 us of idle GPU between the count kernel and the pre-sort."""
    free = _PINNED_FREE.get(n)
    buf = free.pop() if free else torch.empty(n, dtype=torch.int32, pin_memory=True)
    buf.fill_(-1)
    return buf


# (camera, tile) key + depth rank as ONE 32-bit word through emission and pair sort when they fit (gs_isect_finish_presorted's
# n_kept_host): GS_PACKED_PAIRS=0 keeps the (key, flatten id) pairs
_PACKED_PAIRS = os.environ.get("GS_PACKED_PAIRS", "1") != "0"


_WAIT_TIMEOUT_S = float(os.environ.get("GS_WAIT_TIMEOUT_S", "600"))  # backstop of any host wait on a BUSY stream (<= 0: none)
_WAIT_WARNED = [False]


class _SentinelEvent:
    """``query`` / ``synchronize`` of an event over a pinned buffer whose entries go from -1 to >= 0 as the kernel stores them
    (posted 4-byte writes of independent workgroups into host-coherent memory: each becomes visible on its own -- the
    mechanism needs fine-grained coherent pinned memory, HIP's default for ``hipHostMalloc``; with HIP_HOST_COHERENT=0 the
    stores only show at a synchronisation point, which the stream check below turns into a late but correct result).

    The wait is BOUNDED and notices a dead GPU: round 4's query / yield loop for the first few hundred polls, then a yielding spin up to 20 ms, naps after that; every ~2 ms the launch stream is queried -- a
    device fault raises there, and a stream that has drained while the sentinel is still unset means the kernel never stored
    (failed launch, lost write): RuntimeError instead of a core spinning for good.  A stream that is still busy is waited for (one
    warning after 30 s); ``GS_WAIT_TIMEOUT_S`` (600; <= 0: none) is only the backstop behind that."""

    __slots__ = ("buf", "np", "stream", "what")

    def __init__(self, buf: Tensor, stream=None, what: str = "the count kernel's block sums (isect_count_keys_kernel / projection_fwd_kernel)"):
        self.buf = buf
        self.np = buf.numpy()  # (a view of the pinned memory: numpy's min over ~1 K ints is a microsecond, torch's op is ~5)
        self.stream = stream  # the stream the storing kernel was launched on (None: the current one at wait time)
        self.what = what

    def query(self) -> bool:
        a = self.np
        return a[-1] >= 0 and a[0] >= 0 and int(a.min()) >= 0  # (two cache lines while the kernel is far from done)

    def synchronize(self, timeout_s: Optional[float] = None) -> None:
        import time

        query, nap0 = self.query, time.sleep
        # fast phase: exactly round 4's wait (query, yield) for the first few hundred polls -- the usual wait is tens to hundreds of
        # microseconds, up to a step's length when the host runs ahead of the GPU; no clock reads in here (an A/B on one box read
        # 0.744 against 0.738 ms per step with a perf_counter() per poll)
        for _ in range(400 if timeout_s is None else 1):
            if query():
                return
            nap0(0)
        t0 = time.perf_counter()
        limit = _WAIT_TIMEOUT_S if timeout_s is None else timeout_s
        next_check = t0
        while not query():
            now = time.perf_counter()
            if now >= next_check:
                next_check = now + 2e-3
                st = self.stream if self.stream is not None else torch.cuda.current_stream()
                try:
                    drained = st.query()  # raises on a device fault / an earlier HIP error on the stream
                except Exception as e:
                    raise RuntimeError(f"GPU error while waiting for {self.what}: {e}") from e
                if drained:
                    # everything queued has run: stores of a finished kernel are visible now or never
                    if query():
                        return
                    raise RuntimeError(f"the stream drained but {self.what} never arrived in pinned memory "
                                       f"(kernel not launched, faulted, or its stores were lost)")
                # a stream that is still BUSY is not an error: a long evaluation queued ahead, a shared GPU or a profiler serialising
                # kernels can legitimately put many seconds of work in front of the count kernel.  Warn once and keep waiting; the
                # bound (GS_WAIT_TIMEOUT_S, default 600 s; <= 0: none) is a backstop for a hung device whose stream query still answers
                if now - t0 > 30.0 and not _WAIT_WARNED[0]:
                    _WAIT_WARNED[0] = True
                    import warnings

                    warnings.warn(f"gscodec_studio_amd: waited {now - t0:.0f} s for {self.what}; the launch stream is still busy -- waiting on")
                if limit > 0 and now - t0 > limit:
                    raise RuntimeError(f"timed out after {limit:.1f} s (GS_WAIT_TIMEOUT_S) waiting for {self.what}")
            # yielding spin for 20 ms (a thread that napped comes back late: a 50 us time.sleep takes ~100 us on the test hosts, and
            # with naps from 1 ms on a 2-camera step read 2.63 ms instead of 1.37, tools/bench_multicam.py), naps after that: a wait
            # this long is not a step's own
            nap0(0 if now - t0 < 20e-3 else 200e-6)


@torch.no_grad()
def isect_tiles_begin(means2d, radii, depths, tile_size, tile_width, tile_height, sort, C, N, n_elems, camera_ids):
    """First half of ``isect_tiles``: everything up to the data-dependent size -- count, splat-level depth sort,
    prefix sum -- plus an ASYNCHRONOUS read-back of n_isects into pinned memory.  Work launched between
    ``begin`` and ``finish`` (the SH colours in ``rasterization``) runs while the host waits for the count, so the
    GPU does not idle across the one host sync of the pipeline (reference: the blocking ``.item()`` of
    isect_tiles.cu:200)."""
    _require_gpu(means2d, "isect_tiles")
    # means2d may be the first two columns of the splat rows (row stride 16): read in place
    means2d, s_m2 = _row_strided(means2d, 2)
    if s_m2 % 2:
        means2d, s_m2 = means2d.contiguous(), 2
    depths = _f32c(depths)
    radii = radii.contiguous()
    assert radii.dtype == torch.int32, radii.dtype
    dev = means2d.device
    st_ = dict(means2d=means2d, s_m2=s_m2, radii=radii, depths=depths, tile_size=tile_size, tile_width=tile_width,
               tile_height=tile_height, sort=sort, C=C, N=N, n_elems=n_elems, camera_ids=camera_ids, dev=dev)
    st = _stream(means2d)

    n_tiles = tile_width * tile_height
    # the reference computes floor(log2(x)) + 1 in floating point (isect_tiles.cu:155-157)
    tile_n_bits = int(math.floor(math.log2(n_tiles))) + 1 if n_tiles > 0 else 1
    cam_n_bits = int(math.floor(math.log2(C))) + 1 if C > 0 else 1
    assert tile_n_bits + cam_n_bits <= 32, "tile_n_bits + cam_n_bits must be <= 32"
    st_["tile_n_bits"], st_["cam_n_bits"] = tile_n_bits, cam_n_bits

    tiles_per_gauss = torch.empty(radii.shape, dtype=torch.int32, device=dev)
    st_["tiles_per_gauss"] = tiles_per_gauss
    st_["cum"] = st_["perm"] = st_["pinned"] = st_["event"] = st_["n_kept"] = st_["gsums"] = st_["gpre"] = None
    with _device_of(means2d):
        if n_elems > 0:
            if sort:
                # splat-level depth pre-sort: afterwards only the (camera, tile) bits need sorting
                dkeys = torch.empty(n_elems, dtype=torch.int64, device=dev)
                # n_isects = the sum of the per-block counts, known to the host ~100 us of GPU work (pre-sort, prefix sum,
                # SH colours) before the pipeline needs it
                # sum there.  The kernel stores them STRAIGHT into pinned host memory (device-visible under HIP's unified
                # addressing; a few thousand posted 4-byte writes): no device-to-host copy command in the stream.
                n_sums = B.query("gs_isect_count_blocks", n_elems)
                # (a few thousand blocks -- 983 at 1 M splats -- store straight into pinned memory; beyond that the sums are
                # added up on the device and 8 bytes are copied, as in round 1: with 48 K block sums per step at 49 M splats,
                # stored directly OR copied as one 192 KB block, every third or fourth forward stalled the GPU for ~85 ms)
                # every block reports (intersections, visible elements) as one 8-byte store: [n_sums][2]
                direct = n_sums <= _PINNED_DIRECT_MAX
                pinned = _pinned_take(2 * n_sums) if direct else torch.empty(2, dtype=torch.int64, pin_memory=True)
                bsums = pinned if direct else torch.empty(2 * n_sums, dtype=torch.int32, device=dev)
                # the depth pre-sort: up to 2 M elements the BUCKETED form (sampled splitters -> one partition pass -> local
                # sorts in LDS: 4 launches), above that the plain LSD radix sort (4 passes, bandwidth-bound there); the count
                # kernel counts the digits of the first (only) partition pass into the sort's temp buffer either way
                bucketed = _PRESORT["on"] and bool(B.query("gs_presort_applicable", n_elems))
                tb = B.query("gs_presort_temp_bytes" if bucketed else "gs_sort_temp_bytes", n_elems)
                temp = torch.empty(tb, dtype=torch.uint8, device=dev)
                hist_ready = int(B.query("gs_sort_first_hist_applicable", n_elems))
                dvals = None if bucketed else torch.empty(n_elems, dtype=torch.int32, device=dev)  # (bucketed: keys only)
                split = None
                if bucketed:
                    split = presort_split_buffer(dev)
                    B.call("gs_presort_split", n_elems, B.ptr(radii), B.ptr(depths), B.ptr(split), st)
                B.call("gs_isect_count_keys", n_elems, B.ptr(means2d), s_m2, B.ptr(radii), B.ptr(depths), tile_size, tile_width,
                       tile_height, B.ptr(tiles_per_gauss), B.ptr(dkeys), B.ptr(dvals), B.ptr(bsums),
                       B.ptr(temp) if hist_ready else None, tb if hist_ready else 0, B.ptr(split), st)
                if direct:
                    ev = _SentinelEvent(pinned, stream=torch.cuda.current_stream(dev))  # (the stream the count kernel was queued on)
                else:
                    pinned.copy_(bsums.view(-1, 2).sum(0, dtype=torch.int64), non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(torch.cuda.current_stream(dev))
                # culled elements carry the maximal key: the sort drops them in its first pass
                perm = torch.empty(n_elems, dtype=torch.int32, device=dev)
                n_kept = torch.empty(1, dtype=torch.int32, device=dev)
                # the sort's last launch also leaves the tile counts per group of 2^gshift emission positions behind (the block
                # sums of the emission's prefix scan, which gs_isect_emit_presorted then finishes itself: no cumsum launches)
                gshift = int(B.query("gs_isect_emit_group_shift"))
                gsums = torch.empty((n_elems + (1 << gshift) - 1) >> gshift, dtype=torch.int32, device=dev)
                if bucketed:
                    B.call("gs_presort_buckets", n_elems, B.ptr(dkeys), B.ptr(dvals), B.ptr(split), B.ptr(perm), B.ptr(n_kept),
                           B.ptr(temp), tb, B.ptr(tiles_per_gauss), B.ptr(gsums), gshift, _PRESORT["lds_capacity"], st)
                else:
                    ko = torch.empty_like(dkeys)
                    B.call("gs_sort_pairs_u64_i32_drop", n_elems, B.ptr(dkeys), B.ptr(dvals), B.ptr(ko), B.ptr(perm), 32, 64,
                           0x7FFFFFFF, B.ptr(n_kept), B.ptr(temp), tb, hist_ready, B.ptr(tiles_per_gauss), B.ptr(gsums), gshift, st)
                gpre = None
                if gsums.numel() > int(B.query("gs_isect_emit_prefix_from_groups")):  # many groups: one prefix sum over them
                    gpre = torch.empty(gsums.numel(), dtype=torch.int64, device=dev)
                    sb = B.query("gs_cumsum_scratch_bytes", gsums.numel())
                    scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
                    B.call("gs_cumsum_i32", gsums.numel(), B.ptr(gsums), B.ptr(gpre), B.ptr(scratch), sb, st)
                st_["perm"], st_["n_kept"], st_["gsums"], st_["gpre"] = perm, n_kept, gsums, gpre
                st_["sorted_keys"] = dkeys if bucketed else ko  # (depth bits << 32 | element) in perm's order
                cum = None
            else:
                cum = torch.empty(n_elems, dtype=torch.int64, device=dev)
                sb = B.query("gs_cumsum_scratch_bytes", n_elems)
                scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
                B.call("gs_isect_count", n_elems, B.ptr(means2d), s_m2, B.ptr(radii), tile_size, tile_width, tile_height,
                       B.ptr(tiles_per_gauss), st)
                B.call("gs_cumsum_i32", n_elems, B.ptr(tiles_per_gauss), B.ptr(cum), B.ptr(scratch), sb, st)
                pinned = torch.empty(1, dtype=torch.int64, pin_memory=True)
                pinned.copy_(cum[-1:], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(torch.cuda.current_stream(dev))
            st_["cum"], st_["pinned"], st_["event"] = cum, pinned, ev
    return st_


def _wait_event(ev) -> None:
    """Wait for a CUDA event by POLLING it.  ``Event.synchronize()`` spins only briefly and then sleeps; when the GPU needs a
    few milliseconds to get there (49 M splats: 3 ms per forward) the wake-up came ~17 ms late on the bench host -- the
    forward ran at 42 FPS instead of 300.  A few milliseconds of host polling cost nothing here.  Past 0.25 s the wait is
    handed to the event's own ``synchronize`` (a real event sleeps and raises HIP errors; a ``_SentinelEvent`` naps, watches
    the stream for faults and gives up after ``GS_WAIT_TIMEOUT_S``): never an unbounded spin."""
    import time

    if isinstance(ev, _SentinelEvent):
        ev.synchronize()
        return
    deadline = time.perf_counter() + 0.25
    while not ev.query():
        if time.perf_counter() > deadline:  # something long is queued in front: stop burning the core
            ev.synchronize()
            return


def isect_tiles_abandon(st_) -> None:
    """Drop a state ``isect_tiles_begin`` / ``isect_tiles_start`` returned WITHOUT finishing it (the sparse exchange's
    overflow retry, an exception between begin and finish).  The count kernel stores its block sums straight into the
    state's pinned buffer; torch's pinned caching allocator does not track kernel stores, so the buffer must not go back to
    it (or to our free list) before the kernel has run: wait for the state's event first."""
    if st_ is None or st_.get("event") is None:
        return
    _wait_event(st_["event"])
    pinned = st_.get("pinned")
    if pinned is not None and pinned.dtype == torch.int32:
        _PINNED_FREE.setdefault(pinned.numel(), []).append(pinned)
    st_["pinned"] = st_["event"] = None


@torch.no_grad()
def isect_tiles_finish(st_, offsets_for: Optional[int] = None):
    """Second half of ``isect_tiles``: wait for n_isects, emit the (tile, depth) pairs, sort them.
    ``offsets_for`` = n_cameras: also run ``isect_offset_encode`` and return its [C, th, tw] offsets as a fourth value -- on the
    sorted path the three steps are ONE native call (gs_isect_finish_presorted): this is the stretch between the host's
    read-back and the compositing launch, where the host has to stay ahead of the GPU."""
    means2d, radii, depths, dev = st_["means2d"], st_["radii"], st_["depths"], st_["dev"]
    st = _stream(means2d)
    offsets = None
    if offsets_for is not None:  # (allocated before the wait: its size does not depend on the count)
        offsets = torch.empty((offsets_for, st_["tile_height"], st_["tile_width"]), dtype=torch.int32, device=dev)
    n_isects = n_kept = 0
    if st_["event"] is not None:
        _wait_event(st_["event"])  # the one host sync (isect_tiles.cu:200)
        if st_["pinned"].numel() == 1:  # (the unsorted path: cum[-1])
            n_isects = int(st_["pinned"][0])
        else:  # [blocks][2] (or their two totals): intersections, elements the depth pre-sort keeps
            pn = st_["pinned"].numpy()
            pairs = block_sum_totals(pn) if pn.dtype == "int32" and pn.size % 2 == 0 and pn.size >= 2 else pn.reshape(-1, 2).sum(0, dtype="int64")
            n_isects, n_kept = int(pairs[0]), int(pairs[1])
        if st_["pinned"].dtype == torch.int32:
            _PINNED_FREE.setdefault(st_["pinned"].numel(), []).append(st_["pinned"])
            st_["pinned"] = None
    with _device_of(means2d):
        isect_ids = torch.empty(n_isects, dtype=torch.int64, device=dev)
        flatten_ids = torch.empty(n_isects, dtype=torch.int32, device=dev)
        if st_["sort"] and offsets is not None:
            wb = B.query("gs_isect_finish_work_bytes", n_isects)
            work = torch.empty(wb, dtype=torch.uint8, device=dev)
            B.call("gs_isect_finish_presorted", st_["n_elems"], max(st_["N"], 1), n_isects, B.ptr(st_["perm"]), B.ptr(st_["n_kept"]),
                   B.ptr(st_["camera_ids"]), B.ptr(means2d), st_["s_m2"], B.ptr(radii), B.ptr(depths), B.ptr(st_["tiles_per_gauss"]),
                   B.ptr(st_["gsums"]), B.ptr(st_["gpre"]), st_["tile_size"], st_["tile_width"], st_["tile_height"], st_["tile_n_bits"],
                   st_["cam_n_bits"], offsets_for, B.ptr(isect_ids), B.ptr(flatten_ids), B.ptr(offsets), B.ptr(work), wb,
                   n_kept if _PACKED_PAIRS else 0, B.ptr(st_.get("sorted_keys")), st)
            return st_["tiles_per_gauss"], isect_ids, flatten_ids, offsets
        if n_isects > 0 and st_["sort"]:
            # compact form: (32-bit camera|tile key, flatten id) pairs = 8 B instead of 12 through the sort; its last pass
            # writes the reference's 64-bit ids (key << 32 | depth bits) and the flatten ids
            keys32 = torch.empty(n_isects, dtype=torch.int32, device=dev)
            vals = torch.empty(n_isects, dtype=torch.int32, device=dev)
            B.call("gs_isect_emit_presorted", st_["n_elems"], max(st_["N"], 1), B.ptr(st_["perm"]), B.ptr(st_["n_kept"]),
                   B.ptr(st_["camera_ids"]), B.ptr(means2d), st_["s_m2"], B.ptr(radii), B.ptr(depths), B.ptr(st_["tiles_per_gauss"]),
                   B.ptr(st_["gsums"]), B.ptr(st_["gpre"]), st_["tile_size"], st_["tile_width"], st_["tile_height"], st_["tile_n_bits"], 1, None,
                   B.ptr(keys32), B.ptr(vals), st)
            tb = B.query("gs_sort_isect_temp_bytes", n_isects)
            temp = torch.empty(tb, dtype=torch.uint8, device=dev)
            # (only the bits a key can have set take part in the sort: C = 8 is 3 camera bits, not the 4 of the id layout -- two passes
            # instead of three at 1080p; with all 32 bits in use the ids' sign matters to the last pass: left alone)
            key_bits = st_["tile_n_bits"] + st_["cam_n_bits"]
            if key_bits < 32:
                key_bits = max(1, st_["tile_n_bits"] + max(st_["C"] - 1, 0).bit_length())
            B.call("gs_sort_isect_pairs", n_isects, B.ptr(keys32), B.ptr(vals), B.ptr(depths), key_bits,
                   B.ptr(isect_ids), B.ptr(flatten_ids), B.ptr(temp), tb, st)
        elif n_isects > 0:
            B.call("gs_isect_emit", st_["n_elems"], max(st_["N"], 1), B.ptr(st_["perm"]), B.ptr(st_["n_kept"]), B.ptr(st_["camera_ids"]),
                   B.ptr(means2d), st_["s_m2"], B.ptr(radii), B.ptr(depths), B.ptr(st_["cum"]), st_["tile_size"], st_["tile_width"],
                   st_["tile_height"], st_["tile_n_bits"], B.ptr(isect_ids), B.ptr(flatten_ids), st)
        if offsets is not None:
            B.call("gs_isect_offset_encode", n_isects, B.ptr(isect_ids), offsets_for, st_["tile_width"] * st_["tile_height"],
                   st_["tile_n_bits"], B.ptr(offsets), st)
            return st_["tiles_per_gauss"], isect_ids, flatten_ids, offsets
    return st_["tiles_per_gauss"], isect_ids, flatten_ids


@torch.no_grad()
def isect_offset_encode(isect_ids: Tensor, n_cameras: int, tile_width: int, tile_height: int) -> Tensor:
    """Encodes sorted intersection ids to per-(camera, tile) start offsets [C, th, tw] (int32)."""
    _require_gpu(isect_ids, "isect_offset_encode")
    isect_ids = isect_ids.contiguous()
    assert isect_ids.dtype == torch.int64, isect_ids.dtype
    n_isects = isect_ids.shape[0]
    n_tiles = tile_width * tile_height
    tile_n_bits = int(math.floor(math.log2(n_tiles))) + 1 if n_tiles > 0 else 1
    offsets = torch.empty((n_cameras, tile_height, tile_width), dtype=torch.int32, device=isect_ids.device)
    with _device_of(isect_ids):
        B.call("gs_isect_offset_encode", n_isects, B.ptr(isect_ids), n_cameras, n_tiles, tile_n_bits,
               B.ptr(offsets), _stream(isect_ids))
    return offsets


# ---------------------------------------------------------------------------
# rasterize to pixels  (reference _wrapper.py:436-568, 901-1028)
# ---------------------------------------------------------------------------
# Launch tuning of the compositing kernels: (segment length of the backward, solo threshold of the forward, XCD group of the
# forward / backward); -1 = the library's default (the measured MI355X optimum).  The library itself is stateless: the values
# travel in the gs_raster_plan made for every forward and handed to its backward.  Preset from the environment, read once.
_TUNING_KEYS = ("raster_seg", "raster_solo_min", "raster_xcd_fwd", "raster_xcd_bwd", "raster_order_fwd")
_RASTER_TUNING = [int(os.environ.get(k, "-1")) for k in ("GS_RASTER_SEG", "GS_RASTER_SOLO", "GS_RASTER_XCD_FWD", "GS_RASTER_XCD_BWD",
                                                         "GS_RASTER_ORDER_FWD")]


def set_raster_tuning(**kv) -> dict:
    """Set tuning values (keys: raster_seg, raster_solo_min, raster_xcd_fwd, raster_xcd_bwd, raster_order_fwd; ``None`` / -1 = default) for the
    rasterize calls that FOLLOW; returns the previous values.  A forward's values stay with its backward (they are stored
    in its plan), so changing them between the two is harmless."""
    prev = dict(zip(_TUNING_KEYS, _RASTER_TUNING))
    for k, v in kv.items():
        _RASTER_TUNING[_TUNING_KEYS.index(k)] = -1 if v is None else int(v)
    return prev


def _raster_plan(n_tiles_all: int, n_isects: int, channels: int, forward_only: bool = False):
    """(plan buffer, scratch bytes): a gs_raster_plan (host struct, 64 bytes) for one forward / backward pair.
    ``forward_only``: no backward will follow -- segment length 0, i.e. no checkpoints (the scratch then only holds the
    forward's tile order and cost counters)."""
    plan = ctypes.create_string_buffer(64)
    tun = (ctypes.c_int32 * 5)(*_RASTER_TUNING)
    if forward_only:
        tun[0] = 0
    B.call("gs_rasterize_plan", n_tiles_all, n_isects, channels, ctypes.addressof(tun), ctypes.addressof(plan))
    return plan, struct.unpack_from("<Q", plan, 32)[0]


def _splat_layout(means2d: Tensor, conics: Tensor, colors: Tensor, opacities: Tensor):
    """The four per-splat arrays as the kernels take them: contiguous tensors (strides None), or -- column views of wider
    row-major buffers, in particular of the splat rows ``project_rows`` fills -- read IN PLACE through their row strides."""
    channels = colors.shape[-1]
    (m2, s0), (cn, s1), (col, s2), (op, s3) = (_row_strided(means2d, 2), _row_strided(conics, 3), _row_strided(colors, channels),
                                               _elem_strided(opacities))
    if s0 % 2 or m2.data_ptr() % 8:  # (the kernels load a mean as one 8-byte word)
        m2, s0 = m2.contiguous(), 2
    if (s0, s1, s2, s3) == (2, 3, channels, 1):
        return m2, cn, col, op, None
    return m2, cn, col, op, (ctypes.c_uint32 * 4)(s0, s1, s2, s3)


@torch.no_grad()
def _split_big_tiles(isect_offsets: Tensor, flatten_ids: Tensor, masks: Optional[Tensor]):
    """Tiles of 2s x 2s pixels as 2 x 2 sub-tiles of s x s: -> (offsets [C, 2 th, 2 tw], flatten_ids [4 n], masks) where every
    sub-tile owns a copy of its tile's list range (no host synchronisation: the total is 4 n).

    Cost of this detour (tile sizes 18..32 serve no caller in the reference; every trainer uses 16): the list is materialised FOUR
    times (16 n bytes) plus two int32 temporaries of 4 n entries; offsets are int32 like the reference's, so 4 n must stay below 2^31."""
    C, th, tw = isect_offsets.shape
    n = int(flatten_ids.shape[0])
    total = 4 * n
    if total >= 2 ** 31:
        raise RuntimeError(f"tile_size > 16: {n} intersections x 4 sub-tile copies overflow the int32 tile offsets; use tile_size <= 16")
    dev = flatten_ids.device
    i32 = torch.int32
    start = isect_offsets.reshape(-1).to(i32)
    end = torch.cat([start[1:], start.new_full((1,), n)])

    def rep(t):
        return t.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)

    vcount = rep((end - start).view(C, th, tw)).reshape(-1)
    vstart = rep(start.view(C, th, tw)).reshape(-1)
    vend = torch.cumsum(vcount, 0, dtype=i32)  # (< 2^31: checked above)
    voff = vend - vcount
    if total > 0:
        # sub-tile of every output slot by binary search over the running ends (no repeat_interleave + gather temporaries in int64)
        slot = torch.arange(total, device=dev, dtype=i32)
        tile_of = torch.searchsorted(vend, slot, right=True)
        src = (vstart - voff)[tile_of] + slot
        flat = flatten_ids.index_select(0, src)
    else:
        flat = flatten_ids.new_empty(0)
    return voff.view(C, 2 * th, 2 * tw).contiguous(), flat, (rep(masks).contiguous() if masks is not None else None)


def rasterize_to_pixels(
    means2d: Tensor,  # [C, N, 2] or [nnz, 2]
    conics: Tensor,  # [C, N, 3] or [nnz, 3]
    colors: Tensor,  # [C, N, channels] or [nnz, channels]
    opacities: Tensor,  # [C, N] or [nnz]
    image_width: int,
    image_height: int,
    tile_size: int,
    isect_offsets: Tensor,  # [C, tile_height, tile_width]
    flatten_ids: Tensor,  # [n_isects]
    backgrounds: Optional[Tensor] = None,  # [C, channels]
    masks: Optional[Tensor] = None,  # [C, tile_height, tile_width]
    packed: bool = False,
    absgrad: bool = False,
    deterministic: bool = False,
    prefill: Optional["GradPrefill"] = None,
) -> Tuple[Tensor, Tensor]:
    """Rasterizes Gaussians to pixels.

    Returns (render_colors [C,H,W,channels], render_alphas [C,H,W,1]).
    ``deterministic`` (opt-in, not in the reference; up to 4 channels): the backward accumulates the per-splat sums in fixed
    point (integer atomics commute), so two runs give bit-identical gradients; ~60 us slower at 1 M splats.
    """
    C = isect_offsets.size(0)
    if packed:
        nnz = means2d.size(0)
        assert means2d.shape == (nnz, 2), means2d.shape
        assert conics.shape == (nnz, 3), conics.shape
        assert colors.shape[0] == nnz, colors.shape
        assert opacities.shape == (nnz,), opacities.shape
    else:
        N = means2d.size(1)
        assert means2d.shape == (C, N, 2), means2d.shape
        assert conics.shape == (C, N, 3), conics.shape
        assert colors.shape[:2] == (C, N), colors.shape
        assert opacities.shape == (C, N), opacities.shape
    if backgrounds is not None:
        assert backgrounds.shape == (C, colors.shape[-1]), backgrounds.shape
        backgrounds = backgrounds.contiguous()
    if masks is not None:
        assert masks.shape == isect_offsets.shape, masks.shape
        masks = masks.contiguous()

    channels = colors.shape[-1]
    if channels > 513 or channels == 0:
        raise ValueError(f"Unsupported number of color channels: {channels}")
    # NOTE: the reference zero-pads to {1,2,3,4,5,8,9,16,...} channels here because its
    # kernels are template-instantiated per channel count; the HIP ABI takes a runtime
    # channel count, so no padding copy is made.

    tile_height, tile_width = isect_offsets.shape[1:3]
    assert tile_height * tile_size >= image_height, f"Assert Failed: {tile_height} * {tile_size} >= {image_height}"
    assert tile_width * tile_size >= image_width, f"Assert Failed: {tile_width} * {tile_size} >= {image_width}"
    assert 1 <= tile_size <= 32, f"tile_size must be in [1, 32], got {tile_size}"
    if tile_size > 16:
        # The kernels map a tile onto four wave64 quadrants of 8x8 pixels (tiles up to 16x16); the reference launches tile_size^2
        # threads and takes up to 32 (rasterize_to_pixels_fwd.cu:228; no caller of the reference uses more than 16).  A tile of
        # 18, 20, ... 32 pixels is composited as 2 x 2 SUB-TILES of half the size, each walking its own copy of the tile's list
        # (the exact per-quadrant culling drops what a sub-tile does not need): same images and gradients, 4x the list entries.
        if tile_size % 2:
            raise ValueError(f"tile_size {tile_size}: odd tile sizes above 16 are not supported on the HIP backend (1..16 and the even "
                             f"sizes 18..32 are)")
        isect_offsets, flatten_ids, masks = _split_big_tiles(isect_offsets, flatten_ids, masks)
        tile_size //= 2
        tile_height, tile_width = isect_offsets.shape[1:3]

    # (no .contiguous() on the splat arrays: column views of the splat rows are read in place, _splat_layout)
    return _RasterizeToPixels.apply(
        means2d, conics, colors, opacities, backgrounds,
        masks, image_width, image_height, tile_size, isect_offsets.contiguous(), flatten_ids.contiguous(), absgrad,
        bool(deterministic), prefill, torch.is_grad_enabled(),
    )


class _RasterizeToPixels(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, masks, width, height, tile_size,
                isect_offsets, flatten_ids, absgrad, deterministic=False, prefill=None, grad_mode=True):
        _require_gpu(means2d, "rasterize_to_pixels")
        means2d, conics, colors, opacities, strides = _splat_layout(means2d, conics, colors, opacities)
        backgrounds = _f32c(backgrounds)
        C, tile_height, tile_width = isect_offsets.shape
        channels = colors.shape[-1]
        n_elems = opacities.numel()
        n_isects = flatten_ids.shape[0]
        dev = means2d.device
        render_colors = torch.empty((C, height, width, channels), dtype=torch.float32, device=dev)
        render_alphas = torch.empty((C, height, width, 1), dtype=torch.float32, device=dev)
        last_ids = torch.empty((C, height, width), dtype=torch.int32, device=dev)
        m8 = masks.view(torch.uint8) if masks is not None else None
        assert isect_offsets.dtype == torch.int32 and flatten_ids.dtype == torch.int32
        # the scratch carries the forward checkpoints of the depth-segmented backward (128 MB written at 1 M splats /
        # 1080p): only ask for them when a backward can follow
        # (ctx.needs_input_grad says True for parameters even under torch.no_grad(), where no backward can follow: the caller's
        # grad mode comes along as an argument -- inside forward() it always reads False)
        needs_bwd = bool(grad_mode) and any(ctx.needs_input_grad[:5])
        with _device_of(means2d):
            plan, sb = _raster_plan(C * tile_height * tile_width, n_isects, channels, forward_only=not needs_bwd)
            scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
            # the packed gradient rows of the backward ([n_elems,16], accumulated with atomics) are zero-filled by THIS
            # launch, as a side job of the tile workgroups: no fill pass in the backward
            grad_rows = fill = None
            ctx.grad_colors = None
            if needs_bwd and channels <= _FAST_MAX_CHANNELS and n_elems > 0:
                # (+ the per-gaussian gradient tensors the projection node asked for, GradPrefill: one buffer, one fill)
                extra = prefill.floats() if prefill is not None else 0
                if extra:
                    extra += 64  # slack behind the last piece (a multi-GPU reduction rounds the span of all pieces up into it)
                # 5..32 channels: the geometry gradients keep their 16-float rows (the projection backward reads them in
                # place), the colour gradients get a dense [n_elems, channels] array behind them -- same buffer, same fill
                wide = _pad64(n_elems * channels) if channels > 4 else 0
                fill = torch.empty(n_elems * 16 + wide + extra, dtype=torch.float32, device=dev)
                grad_rows = fill[:n_elems * 16].view(opacities.shape + (16,))
                if wide:
                    ctx.grad_colors = fill[n_elems * 16:n_elems * 16 + n_elems * channels].view(opacities.shape + (channels,))
                if extra:
                    prefill.carve(fill, n_elems * 16 + wide)
            B.call("gs_rasterize_fwd", C, n_elems, n_isects, channels, B.ptr(means2d), B.ptr(conics), B.ptr(colors),
                   B.ptr(opacities), ctypes.addressof(strides) if strides is not None else None, B.ptr(backgrounds), B.ptr(m8),
                   width, height, tile_size, tile_width,
                   tile_height, B.ptr(isect_offsets), B.ptr(flatten_ids), B.ptr(render_colors),
                   B.ptr(render_alphas), B.ptr(last_ids), ctypes.addressof(plan) if plan is not None else None,
                   B.ptr(scratch) if plan is not None else None,
                   B.ptr(fill), fill.numel() * 4 if fill is not None else 0, _stream(means2d))
        ctx.grad_rows = grad_rows  # consumed by the first backward; a repeated one (retain_graph) fills its own
        ctx.plan, ctx.strides = plan, strides  # host structs: the backward runs under the forward's plan and layout
        # scratch carries the forward checkpoints of the depth-segmented backward.  The segmented backward rebuilds
        # "colour behind the segment" from the FINAL render (B = v_out . (colour_final - colour_ckpt)), so the output is
        # saved through save_for_backward: autograd then version-checks it and an in-place edit of the returned image
        # before backward() raises instead of silently corrupting the gradients (the reference does not need its
        # output in the backward, so this is the one place where in-place post-processing must become out-of-place).
        ctx.save_for_backward(means2d, conics, colors, opacities, backgrounds, masks, isect_offsets, flatten_ids,
                              render_alphas, last_ids, scratch, render_colors)
        ctx.width, ctx.height, ctx.tile_size, ctx.absgrad = width, height, tile_size, absgrad
        if deterministic and channels > 4:
            raise RuntimeError("rasterize_to_pixels(deterministic=True) supports up to 4 channels")
        ctx.deterministic = bool(deterministic)
        ctx.set_materialize_grads(False)
        return render_colors, render_alphas

    @staticmethod
    def backward(ctx, v_render_colors: Tensor, v_render_alphas: Tensor):
        (means2d, conics, colors, opacities, backgrounds, masks, isect_offsets, flatten_ids, render_alphas,
         last_ids, scratch, render_colors) = ctx.saved_tensors
        C, tile_height, tile_width = isect_offsets.shape
        channels = colors.shape[-1]
        n_elems = opacities.numel()
        n_isects = flatten_ids.shape[0]
        # undefined upstream gradients arrive as None (set_materialize_grads(False) in forward): the kernel takes a
        # NULL v_render_alphas, which saves a [C,H,W] zero-fill and one of the 12 per-pixel loads of every work item
        if v_render_colors is None:
            v_render_colors = torch.zeros_like(render_colors)
        v_render_colors, vrc_pix, vrc_ch = _pixel_strided(v_render_colors)
        v_render_alphas = _f32c(v_render_alphas) if v_render_alphas is not None else None
        # accumulated with atomics -> zero-filled.  Up to 4 channels: ONE packed [n_elems,16] buffer
        # (64-byte row per splat: vx vy | ca cb cc | o | c0..c3 | ax ay) so that a splat's whole
        # gradient is one L2 request; the tensors handed to autograd are views of it.
        packed = channels <= _FAST_MAX_CHANNELS
        # deterministic mode: fixed-point sums in an int64 buffer of their own; the float rows are then WRITTEN by a second kernel
        det = torch.zeros((n_elems, 2, 12), dtype=torch.int64, device=means2d.device) if (ctx.deterministic and n_elems > 0) else None
        if packed:
            P, ctx.grad_rows = ctx.grad_rows, None
            if P is None:
                # (the deterministic route's finalize kernel WRITES every row -- except when there is nothing to composite:
                # gs_rasterize_bwd returns before it with n_isects == 0, and the rows must then be zeros, not stale memory)
                P = (torch.empty if (det is not None and n_isects > 0) else torch.zeros)(
                    opacities.shape + (16,), dtype=torch.float32, device=means2d.device)
            v_means2d, v_conics, v_opacities = P[..., 0:2], P[..., 2:5], P[..., 5]
            v_means2d_abs = P[..., 10:12] if ctx.absgrad else None
            if channels <= 4:
                v_colors = P[..., 6:6 + channels]
                out_ptrs = (B.ptr(P) if ctx.absgrad else None, B.ptr(P), None, None, None)
            else:  # geometry rows + the colour gradients in their own dense array (packed16 = 2)
                v_colors, ctx.grad_colors = ctx.grad_colors, None
                if v_colors is None:
                    v_colors = torch.zeros(opacities.shape + (channels,), dtype=torch.float32, device=means2d.device)
                out_ptrs = (B.ptr(P) if ctx.absgrad else None, B.ptr(P), None, B.ptr(v_colors), None)
        else:
            v_means2d = torch.zeros_like(means2d)
            v_conics = torch.zeros_like(conics)
            v_colors = torch.zeros_like(colors)
            v_opacities = torch.zeros_like(opacities)
            v_means2d_abs = torch.zeros_like(means2d) if ctx.absgrad else None
            out_ptrs = (B.ptr(v_means2d_abs), B.ptr(v_means2d), B.ptr(v_conics), B.ptr(v_colors), B.ptr(v_opacities))
        m8 = masks.view(torch.uint8) if masks is not None else None
        plan, strides = ctx.plan, ctx.strides
        with _device_of(means2d):
            B.call("gs_rasterize_bwd", C, n_elems, n_isects, channels, B.ptr(means2d), B.ptr(conics), B.ptr(colors),
                   B.ptr(opacities), ctypes.addressof(strides) if strides is not None else None, B.ptr(backgrounds), B.ptr(m8),
                   ctx.width, ctx.height, ctx.tile_size,
                   tile_width, tile_height, B.ptr(isect_offsets), B.ptr(flatten_ids), B.ptr(render_colors),
                   B.ptr(render_alphas), B.ptr(last_ids), B.ptr(v_render_colors), B.ptr(v_render_alphas), vrc_pix, vrc_ch, *out_ptrs,
                   (1 if channels <= 4 else 2) if packed else 0, B.ptr(det), ctypes.addressof(plan) if plan is not None else None,
                   B.ptr(scratch) if plan is not None else None, _stream(means2d))
        if ctx.absgrad:
            means2d.absgrad = v_means2d_abs
        if ctx.needs_input_grad[4]:
            v_backgrounds = (v_render_colors * (1.0 - render_alphas).float()).sum(dim=(1, 2))
        else:
            v_backgrounds = None
        return (v_means2d, v_conics, v_colors, v_opacities, v_backgrounds) + (None,) * 10
