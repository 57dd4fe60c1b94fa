"""numpy/ctypes front-end of oracle/gs_oracle.c (TEST INFRASTRUCTURE ONLY).

Every function takes and returns numpy arrays (fp32 / int32 / int64 / uint8) and mirrors
one stage of the reference's path; see the header of gs_oracle.c for the file:line map.
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import Optional, Tuple

import numpy as np

_DIR = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_DIR, "_build", "libgs_oracle.so")
_LIB64_PATH = os.path.join(_DIR, "_build", "libgs_oracle_f64.so")
_lib = None
_lib64 = None
_BITS = [32]


class precision:
    """``with precision(64):`` -- run the oracle functions of this module in float64: the SAME C source compiled with
    every ``float`` replaced by ``double`` (oracle/Makefile, libgs_oracle_f64.so), numpy arrays in and out as float64.
    The threshold constants (1/255, 0.999, 1e-4) keep their fp32 values, so it is the reference's algorithm evaluated
    without rounding noise: the ground truth the gradient tests measure both the HIP kernels and the fp32 oracle against.
    Only the floating-point stages (projection, SH, compositing) are meaningful in this mode."""

    def __init__(self, bits: int):
        assert bits in (32, 64)
        self.bits = bits

    def __enter__(self):
        _BITS.append(self.bits)
        return self

    def __exit__(self, *a):
        _BITS.pop()


def _real():
    return np.float64 if _BITS[-1] == 64 else np.float32

CAMERA_MODELS = {"pinhole": 0, "ortho": 1, "fisheye": 2}


def build(force: bool = False) -> str:
    src = os.path.join(_DIR, "gs_oracle.c")
    stale = any(not os.path.exists(q) or os.path.getmtime(q) < os.path.getmtime(src) for q in (_LIB_PATH, _LIB64_PATH))
    if force or stale:
        subprocess.check_call(["make", "-C", _DIR] + (["-B"] if force else ["-s"]))
    return _LIB_PATH


def lib():
    global _lib, _lib64
    if _BITS[-1] == 64:
        if _lib64 is None:
            if not os.path.exists(_LIB64_PATH):
                build()
            _lib64 = ctypes.CDLL(_LIB64_PATH)
            _lib64.orc_isect_count.restype = ctypes.c_int64
        return _lib64
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.orc_isect_count.restype = ctypes.c_int64
    return _lib


def _p(a: Optional[np.ndarray]):
    if a is None:
        return ctypes.c_void_p(None)
    assert a.flags["C_CONTIGUOUS"], "oracle arrays must be contiguous"
    return ctypes.c_void_p(a.ctypes.data)


def _f(a) -> Optional[np.ndarray]:
    return None if a is None else np.ascontiguousarray(a, dtype=_real())


_u32, _u64, _i32 = ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int32


def _cf(v):
    return ctypes.c_double(v) if _BITS[-1] == 64 else ctypes.c_float(v)


def projection_fwd(means, covars, quats, scales, viewmats, Ks, width, height, eps2d=0.3, near_plane=0.01,
                   far_plane=1e10, radius_clip=0.0, calc_compensations=False, camera_model="pinhole",
                   packed_formula=False):
    """-> radii [C,N] i32, means2d [C,N,2], depths [C,N], conics [C,N,3], compensations [C,N] | None.
    Entries with radii == 0 hold zeros (the reference leaves them uninitialised)."""
    means, covars, quats, scales, viewmats, Ks = map(_f, (means, covars, quats, scales, viewmats, Ks))
    C, N = viewmats.shape[0], means.shape[0]
    radii = np.zeros((C, N), np.int32)
    means2d = np.zeros((C, N, 2), _real())
    depths = np.zeros((C, N), _real())
    conics = np.zeros((C, N, 3), _real())
    comp = np.zeros((C, N), _real()) if calc_compensations else None
    lib().orc_projection_fwd(_u32(C), _u32(N), _p(means), _p(covars), _p(quats), _p(scales), _p(viewmats), _p(Ks),
                             _i32(width), _i32(height), _cf(eps2d), _cf(near_plane), _cf(far_plane),
                             _cf(radius_clip), _i32(CAMERA_MODELS[camera_model]), _i32(int(packed_formula)),
                             _p(radii), _p(means2d), _p(depths), _p(conics), _p(comp))
    return radii, means2d, depths, conics, comp


def projection_bwd(means, covars, quats, scales, viewmats, Ks, width, height, eps2d, camera_model, radii, conics,
                   compensations, v_means2d, v_depths, v_conics, v_compensations, need_viewmats=True):
    """-> v_means [N,3], v_covars [N,6] | None, v_quats [N,4] | None, v_scales [N,3] | None, v_viewmats [C,4,4] | None"""
    means, covars, quats, scales, viewmats, Ks = map(_f, (means, covars, quats, scales, viewmats, Ks))
    conics, compensations = _f(conics), _f(compensations)
    v_means2d, v_depths, v_conics, v_compensations = map(_f, (v_means2d, v_depths, v_conics, v_compensations))
    radii = np.ascontiguousarray(radii, np.int32)
    C, N = viewmats.shape[0], means.shape[0]
    v_means = np.zeros((N, 3), _real())
    v_covars = np.zeros((N, 6), _real()) if covars is not None else None
    v_quats = np.zeros((N, 4), _real()) if covars is None else None
    v_scales = np.zeros((N, 3), _real()) if covars is None else None
    v_view = np.zeros((C, 4, 4), _real()) if need_viewmats else None
    lib().orc_projection_bwd(_u32(C), _u32(N), _p(means), _p(covars), _p(quats), _p(scales), _p(viewmats), _p(Ks),
                             _i32(width), _i32(height), _cf(eps2d), _i32(CAMERA_MODELS[camera_model]), _p(radii),
                             _p(conics), _p(compensations), _p(v_means2d), _p(v_depths), _p(v_conics),
                             _p(v_compensations), _p(v_means), _p(v_covars), _p(v_quats), _p(v_scales), _p(v_view))
    return v_means, v_covars, v_quats, v_scales, v_view


def sh_fwd(degree, dirs, coeffs, masks=None):
    """dirs [...,3], coeffs [...,K,3] -> colors [...,3] (zeros where masked out)."""
    dirs, coeffs = _f(dirs), _f(coeffs)
    K = coeffs.shape[-2]
    n = dirs.size // 3
    colors = np.zeros_like(dirs)
    m = None if masks is None else np.ascontiguousarray(masks, np.uint8)
    lib().orc_sh_fwd(_u64(n), _u32(K), _u32(degree), _p(dirs), _p(coeffs), _p(m), _p(colors))
    return colors


def sh_bwd(degree, dirs, coeffs, v_colors, masks=None, compute_v_dirs=True):
    dirs, coeffs, v_colors = _f(dirs), _f(coeffs), _f(v_colors)
    K = coeffs.shape[-2]
    n = dirs.size // 3
    v_coeffs = np.zeros_like(coeffs)
    v_dirs = np.zeros_like(dirs) if compute_v_dirs else None
    m = None if masks is None else np.ascontiguousarray(masks, np.uint8)
    lib().orc_sh_bwd(_u64(n), _u32(K), _u32(degree), _p(dirs), _p(coeffs), _p(m), _p(v_colors), _p(v_coeffs),
                     _p(v_dirs))
    return v_coeffs, v_dirs


def tile_bits(n_tiles: int, n_cameras: int) -> Tuple[int, int]:
    """floor(log2(x)) + 1, as isect_tiles.cu:155-157"""
    return int(math.floor(math.log2(n_tiles))) + 1, int(math.floor(math.log2(n_cameras))) + 1


def isect_tiles(means2d, radii, depths, tile_size, tile_width, tile_height, sort=True, n_cameras=None,
                camera_ids=None):
    """-> tiles_per_gauss (shape of radii, i32), isect_ids [I] i64, flatten_ids [I] i32"""
    means2d, depths = _f(means2d), _f(depths)
    radii = np.ascontiguousarray(radii, np.int32)
    n_elems = radii.size
    if camera_ids is None:
        C, N = radii.shape
    else:
        C, N = n_cameras, 1
        camera_ids = np.ascontiguousarray(camera_ids, np.int64)
    tpg = np.zeros(radii.shape, np.int32)
    n_isects = lib().orc_isect_count(_u64(n_elems), _p(means2d), _p(radii), _u32(tile_size), _u32(tile_width),
                                     _u32(tile_height), _p(tpg))
    tb, cb = tile_bits(tile_width * tile_height, C)
    ids = np.zeros(n_isects, np.int64)
    flat = np.zeros(n_isects, np.int32)
    lib().orc_isect_emit(_u64(n_elems), _u32(max(N, 1)), _p(camera_ids), _p(means2d), _p(radii), _p(depths),
                         _u32(tile_size), _u32(tile_width), _u32(tile_height), _u32(tb), _p(ids), _p(flat))
    if sort and n_isects:
        lib().orc_sort_pairs(_u64(n_isects), _p(ids), _p(flat), _i32(32 + tb + cb))
    return tpg, ids, flat


def sort_pairs(keys, vals, end_bit):
    keys = np.array(keys, dtype=np.int64, copy=True)
    vals = np.array(vals, dtype=np.int32, copy=True)
    lib().orc_sort_pairs(_u64(keys.size), _p(keys), _p(vals), _i32(end_bit))
    return keys, vals


def isect_offset_encode(isect_ids, n_cameras, tile_width, tile_height):
    isect_ids = np.ascontiguousarray(isect_ids, np.int64)
    n_tiles = tile_width * tile_height
    tb, _ = tile_bits(n_tiles, n_cameras)
    offsets = np.zeros((n_cameras, tile_height, tile_width), np.int32)
    lib().orc_isect_offset_encode(_u64(isect_ids.size), _p(isect_ids), _u32(n_cameras), _u32(n_tiles), _u32(tb),
                                  _p(offsets))
    return offsets


def rasterize_fwd(means2d, conics, colors, opacities, width, height, tile_size, isect_offsets, flatten_ids,
                  backgrounds=None, masks=None, return_borderline=False):
    """-> render_colors [C,H,W,D], render_alphas [C,H,W,1], last_ids [C,H,W] (+ borderline [C,H,W] u8)"""
    means2d, conics, colors, opacities, backgrounds = map(_f, (means2d, conics, colors, opacities, backgrounds))
    isect_offsets = np.ascontiguousarray(isect_offsets, np.int32)
    flatten_ids = np.ascontiguousarray(flatten_ids, np.int32)
    C, th, tw = isect_offsets.shape
    D = colors.shape[-1]
    rc = np.zeros((C, height, width, D), _real())
    ra = np.zeros((C, height, width, 1), _real())
    li = np.zeros((C, height, width), np.int32)
    bl = np.zeros((C, height, width), np.uint8)
    m = None if masks is None else np.ascontiguousarray(masks, np.uint8)
    lib().orc_rasterize_fwd(_u32(C), _u32(flatten_ids.size), _u32(D), _p(means2d), _p(conics), _p(colors),
                            _p(opacities), _p(backgrounds), _p(m), _u32(width), _u32(height), _u32(tile_size),
                            _u32(tw), _u32(th), _p(isect_offsets), _p(flatten_ids), _p(rc), _p(ra), _p(li), _p(bl))
    if return_borderline:
        return rc, ra, li, bl
    return rc, ra, li


def rasterize_max_weight(means2d, conics, opacities, width, height, tile_size, isect_offsets, flatten_ids):
    """-> [C,H,W]: the largest blending weight alpha * T any single list entry could contribute to the pixel (thresholds
    ignored): a flipped threshold decision moves the pixel's colour by at most 2 x this x max|colour| (gs_oracle.c)."""
    means2d, conics, opacities = map(_f, (means2d, conics, opacities))
    isect_offsets = np.ascontiguousarray(isect_offsets, np.int32)
    flatten_ids = np.ascontiguousarray(flatten_ids, np.int32)
    C, th, tw = isect_offsets.shape
    mw = np.zeros((C, height, width), _real())
    lib().orc_rasterize_max_weight(_u32(C), _u32(flatten_ids.size), _p(means2d), _p(conics), _p(opacities), _u32(width),
                                   _u32(height), _u32(tile_size), _u32(tw), _u32(th), _p(isect_offsets), _p(flatten_ids), _p(mw))
    return mw


def rasterize_bwd(means2d, conics, colors, opacities, width, height, tile_size, isect_offsets, flatten_ids,
                  render_alphas, last_ids, v_render_colors, v_render_alphas, backgrounds=None, masks=None,
                  absgrad=False):
    """-> v_means2d, v_conics, v_colors, v_opacities, v_means2d_abs | None (shaped like the inputs)"""
    means2d, conics, colors, opacities, backgrounds = map(_f, (means2d, conics, colors, opacities, backgrounds))
    render_alphas, v_render_colors, v_render_alphas = map(_f, (render_alphas, v_render_colors, v_render_alphas))
    isect_offsets = np.ascontiguousarray(isect_offsets, np.int32)
    flatten_ids = np.ascontiguousarray(flatten_ids, np.int32)
    last_ids = np.ascontiguousarray(last_ids, np.int32)
    C, th, tw = isect_offsets.shape
    D = colors.shape[-1]
    n_elems = opacities.size
    v_m = np.zeros_like(means2d)
    v_c = np.zeros_like(conics)
    v_col = np.zeros_like(colors)
    v_o = np.zeros_like(opacities)
    v_abs = np.zeros_like(means2d) if absgrad else None
    m = None if masks is None else np.ascontiguousarray(masks, np.uint8)
    lib().orc_rasterize_bwd(_u32(C), _u32(n_elems), _u32(flatten_ids.size), _u32(D), _p(means2d), _p(conics),
                            _p(colors), _p(opacities), _p(backgrounds), _p(m), _u32(width), _u32(height),
                            _u32(tile_size), _u32(tw), _u32(th), _p(isect_offsets), _p(flatten_ids),
                            _p(render_alphas), _p(last_ids), _p(v_render_colors), _p(v_render_alphas), _p(v_abs),
                            _p(v_m), _p(v_c), _p(v_col), _p(v_o))
    return v_m, v_c, v_col, v_o, v_abs


def f32(v: float) -> float:
    return float(np.float32(v))


def quant_noise_fwd(x, noise, lo, hi, q_step):
    x, noise = _f(x), _f(noise)
    out = np.zeros_like(x)
    lib().orc_quant_noise_fwd(_u64(x.size), _p(x), _p(noise), _cf(f32(lo)), _cf(f32(hi)), _cf(f32(q_step)), _p(out))
    return out


def quant_noise_bwd(x, v_out, lo, hi):
    x, v_out = _f(x), _f(v_out)
    v_x = np.zeros_like(x)
    lib().orc_quant_noise_bwd(_u64(x.size), _p(x), _p(v_out), _cf(f32(lo)), _cf(f32(hi)), _p(v_x))
    return v_x


def quant_round_fwd(x, lo, hi, bitwidth):
    """-> (clamped x, output); mirrors STE.forward incl. the in-place clamp (returned as a copy)."""
    x = np.array(x, dtype=np.float32, copy=True)
    out = np.zeros_like(x)
    lib().orc_quant_round_fwd(_u64(x.size), _p(x), _cf(f32(lo)), _cf(f32(hi)), _cf(f32(hi - lo)),
                              _cf(f32(1 / (2**bitwidth - 1))), _p(out))
    return x, out


def rasterization(means, quats, scales, opacities, colors, viewmats, Ks, width, height, near_plane=0.01,
                  far_plane=1e10, radius_clip=0.0, eps2d=0.3, sh_degree=None, tile_size=16, backgrounds=None,
                  camera_model="pinhole", antialiased=False):
    """Whole unpacked forward path (projection -> SH -> isect/sort -> offsets -> compositing), as
    gsplat/rendering.py:28-582 orchestrates it.  Returns (render_colors, render_alphas, meta dict)."""
    means, quats, scales, opacities, colors, viewmats, Ks = map(_f, (means, quats, scales, opacities, colors, viewmats, Ks))
    C, N = viewmats.shape[0], means.shape[0]
    radii, means2d, depths, conics, comp = projection_fwd(
        means, None, quats, scales, viewmats, Ks, width, height, eps2d, near_plane, far_plane, radius_clip,
        calc_compensations=antialiased, camera_model=camera_model)
    opac = np.ascontiguousarray(np.broadcast_to(opacities[None], (C, N)))
    if comp is not None:
        opac = opac * comp
    if sh_degree is None:
        cols = np.ascontiguousarray(np.broadcast_to(colors[None], (C,) + colors.shape)) if colors.ndim == 2 else colors
    else:
        c2w = np.linalg.inv(viewmats.astype(np.float64)).astype(_real())
        dirs = means[None] - c2w[:, None, :3, 3]
        shs = np.ascontiguousarray(np.broadcast_to(colors[None], (C,) + colors.shape)) if colors.ndim == 3 else colors
        cols = sh_fwd(sh_degree, dirs, shs, masks=radii > 0)
        cols = np.maximum(cols + 0.5, 0.0).astype(_real())
    tw, th = math.ceil(width / tile_size), math.ceil(height / tile_size)
    tpg, ids, flat = isect_tiles(means2d, radii, depths, tile_size, tw, th)
    offs = isect_offset_encode(ids, C, tw, th)
    rc, ra, li = rasterize_fwd(means2d, conics, cols, opac, width, height, tile_size, offs, flat, backgrounds)
    meta = dict(radii=radii, means2d=means2d, depths=depths, conics=conics, opacities=opac, colors=cols,
                tiles_per_gauss=tpg, isect_ids=ids, flatten_ids=flat, isect_offsets=offs, last_ids=li,
                tile_width=tw, tile_height=th)
    return rc, ra, meta
