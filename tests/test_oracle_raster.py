"""CPU: the oracle's compositing (the one stage with no runnable reference here) against an
independent dense float64 torch formulation with autograd, and against finite differences.
Constants pinned: alpha clamp 0.999, alpha skip 1/255, exclusive stop at T <= 1e-4, last_ids."""
import math

import numpy as np
import pytest
import torch

from oracle import gs_oracle as O
from util import assert_close, rel_l2


def dense_composite(means2d, conics, colors, opac, order, W, H, bg=None):
    """All pixels x all splats (one camera, one list `order` = depth-sorted splat ids), float64."""
    ys, xs = torch.meshgrid(torch.arange(H, dtype=torch.float64) + 0.5, torch.arange(W, dtype=torch.float64) + 0.5, indexing="ij")
    px, py = xs.reshape(-1, 1), ys.reshape(-1, 1)
    m, c, col, o = means2d[order], conics[order], colors[order], opac[order]
    dx, dy = m[None, :, 0] - px, m[None, :, 1] - py
    sigma = 0.5 * (c[None, :, 0] * dx * dx + c[None, :, 2] * dy * dy) + c[None, :, 1] * dx * dy
    alpha = torch.clamp_max(o[None] * torch.exp(-sigma), 0.999)
    keep = (sigma >= 0) & (alpha >= 1.0 / 255.0)
    alpha = torch.where(keep, alpha, torch.zeros_like(alpha))
    T_after = torch.cumprod(1 - alpha, dim=1)
    T_before = torch.cat([torch.ones_like(T_after[:, :1]), T_after[:, :-1]], dim=1)
    stop = (keep & (T_after <= 1e-4)).to(torch.float64)
    stopped = (torch.cumsum(stop, dim=1) > 0)  # the stopping splat itself is excluded
    w = torch.where(stopped, torch.zeros_like(alpha), alpha * T_before)
    rgb = w @ col
    T_final = 1 - w.sum(1)  # == product over composited splats
    # exact T_final: product of (1 - alpha) over non-stopped
    T_final = torch.prod(torch.where(stopped, torch.ones_like(alpha), 1 - alpha), dim=1)
    if bg is not None:
        rgb = rgb + T_final[:, None] * bg[None]
    contributes = (w > 0)
    last = torch.where(contributes.any(1), (contributes * torch.arange(1, len(order) + 1)).max(1).values - 1, torch.zeros(1, dtype=torch.long))
    return rgb.reshape(H, W, -1), (1 - T_final).reshape(H, W, 1), last.reshape(H, W)


def make_scene(n=120, W=16, H=16, seed=0, opaque=False):
    rs = np.random.RandomState(seed)
    means2d = (rs.rand(1, n, 2) * [W + 8, H + 8] - 4).astype(np.float32)
    s = rs.rand(n) * 3 + 1.0
    th = rs.rand(n) * np.pi
    ax, ay = s, s * (0.3 + rs.rand(n))
    R = np.stack([np.cos(th), -np.sin(th), np.sin(th), np.cos(th)], -1).reshape(n, 2, 2)
    S = np.einsum("nij,nj,nkj->nik", R, np.stack([ax**2, ay**2], -1), R)
    Si = np.linalg.inv(S)
    conics = np.stack([Si[:, 0, 0], Si[:, 0, 1], Si[:, 1, 1]], -1)[None].astype(np.float32)
    opac = (rs.rand(1, n) * (0.6 if not opaque else 0.2) + (0.0 if not opaque else 0.85)).astype(np.float32)
    colors = rs.rand(1, n, 3).astype(np.float32)
    depths = (rs.rand(1, n) + 0.5).astype(np.float32)
    radii = np.ceil(3 * np.sqrt(np.maximum(ax, ay) ** 2)).astype(np.int32)[None]
    return means2d, conics, colors, opac, depths, radii


@pytest.mark.parametrize("opaque", [False, True])
def test_forward_backward_vs_dense_autograd(opaque):
    W = H = 16
    means2d, conics, colors, opac, depths, radii = make_scene(opaque=opaque, n=150)
    tpg, ids, flat = O.isect_tiles(means2d, radii + 40, depths, 16, 1, 1)  # everything in the single tile
    offs = O.isect_offset_encode(ids, 1, 1, 1)
    bg = np.array([[0.3, 0.6, 0.1]], np.float32)
    rc, ra, li, bl = O.rasterize_fwd(means2d, conics, colors, opac, W, H, 16, offs, flat, backgrounds=bg, return_borderline=True)
    t = lambda a: torch.tensor(a[0], dtype=torch.float64, requires_grad=True)
    m_t, c_t, col_t, o_t = t(means2d), t(conics), t(colors), t(opac)
    d_rgb, d_a, d_last = dense_composite(m_t, c_t, col_t, o_t, torch.tensor(flat.astype(np.int64)), W, H, torch.tensor(bg[0], dtype=torch.float64))
    ok = bl[0] == 0
    assert ok.mean() > 0.98
    assert_close(rc[0][ok], d_rgb.detach().numpy()[ok], 1e-4, 1e-5, "colors")
    assert_close(ra[0][ok], d_a.detach().numpy()[ok], 1e-4, 1e-5, "alphas")
    if opaque:
        assert (ra[0] > 0.9998).any(), "the opaque scene must exercise early termination"
    # last_ids: index (in the sorted list) of the last composited splat
    assert np.array_equal(li[0][ok], d_last.numpy()[ok])
    rs = np.random.RandomState(1)
    v_rc = rs.randn(1, H, W, 3).astype(np.float32) * ok[None, ..., None]
    v_ra = rs.randn(1, H, W, 1).astype(np.float32) * ok[None, ..., None]
    loss = (d_rgb * torch.tensor(v_rc[0], dtype=torch.float64)).sum() + (d_a * torch.tensor(v_ra[0], dtype=torch.float64)).sum()
    g_m, g_c, g_col, g_o = torch.autograd.grad(loss, (m_t, c_t, col_t, o_t))
    v_m, v_c, v_col, v_o, v_abs = O.rasterize_bwd(means2d, conics, colors, opac, W, H, 16, offs, flat, ra, li, v_rc, v_ra,
                                                  backgrounds=bg, absgrad=True)
    for name, got, ref in (("v_means2d", v_m[0], g_m), ("v_conics", v_c[0], g_c), ("v_colors", v_col[0], g_col), ("v_opacities", v_o[0], g_o)):
        assert rel_l2(got, ref.numpy()) < 3e-4, (name, rel_l2(got, ref.numpy()))
    assert (v_abs >= np.abs(v_m) - 1e-4).all()


def test_backward_finite_differences():
    """Finite differences w.r.t. the colours (the render is exactly linear in them, so FD is exact
    up to rounding; FD w.r.t. opacity / geometry is polluted by the alpha >= 1/255 threshold ring)."""
    W = H = 16
    means2d, conics, colors, opac, depths, radii = make_scene(n=40, seed=3)
    tpg, ids, flat = O.isect_tiles(means2d, radii + 40, depths, 16, 1, 1)
    offs = O.isect_offset_encode(ids, 1, 1, 1)
    rs = np.random.RandomState(2)
    v_rc = rs.randn(1, H, W, 3).astype(np.float32)

    def f(c):
        rc, ra, li = O.rasterize_fwd(means2d, conics, c, opac, W, H, 16, offs, flat)
        return float((rc.astype(np.float64) * v_rc).sum()), ra, li

    base, ra, li = f(colors)
    _, _, v_col, _, _ = O.rasterize_bwd(means2d, conics, colors, opac, W, H, 16, offs, flat, ra, li, v_rc, np.zeros_like(ra))
    checked = 0
    for j in range(0, 40, 3):
        for k in range(3):
            h = 0.25
            c2 = colors.copy(); c2[0, j, k] += h
            c3 = colors.copy(); c3[0, j, k] -= h
            fd = (f(c2)[0] - f(c3)[0]) / (2 * h)
            assert abs(fd - v_col[0, j, k]) < 1e-3 * max(1.0, abs(fd)), (j, k, fd, v_col[0, j, k])
            checked += abs(fd) > 1e-3
    assert checked >= 10


def test_masks_multi_tile_and_empty():
    means2d, conics, colors, opac, depths, radii = make_scene(n=60, W=40, H=24, seed=5)
    W, H, ts = 40, 24, 16
    tw, th = math.ceil(W / ts), math.ceil(H / ts)
    tpg, ids, flat = O.isect_tiles(means2d, radii, depths, ts, tw, th)
    offs = O.isect_offset_encode(ids, 1, tw, th)
    masks = np.array([[[True, False, True], [False, True, True]]])
    rc, ra, li = O.rasterize_fwd(means2d, conics, colors, opac, W, H, ts, offs, flat, masks=masks)
    rc0, ra0, li0 = O.rasterize_fwd(means2d, conics, colors, opac, W, H, ts, offs, flat)
    pm = np.repeat(np.repeat(masks, ts, 1), ts, 2)[:, :H, :W]
    assert np.array_equal(rc[pm], rc0[pm]) and (rc[~pm] == 0).all()
    # no intersections at all
    z = np.zeros((0,), np.int32)
    rc, ra, li = O.rasterize_fwd(means2d, conics, colors, opac, W, H, ts, np.zeros((1, th, tw), np.int32), z)
    assert (rc == 0).all() and (ra == 0).all() and (li == 0).all()


def test_float64_build_matches_dense_autograd_to_rounding():
    """The float64 build of the oracle (same C source, `float` -> `double`) is the ground truth of the GPU gradient tests:
    here it is pinned against the independent dense float64 torch formulation to 1e-7 (forward and all four gradients;
    not tighter because the alpha clamp is the fp32 VALUE of 0.999 in the oracle and the double 0.999 in the dense formulation)."""
    W = H = 16
    means2d, conics, colors, opac, depths, radii = make_scene(opaque=True, n=150, seed=7)
    tpg, ids, flat = O.isect_tiles(means2d, radii + 40, depths, 16, 1, 1)
    offs = O.isect_offset_encode(ids, 1, 1, 1)
    bg = np.array([[0.3, 0.6, 0.1]], np.float32)
    with O.precision(64):
        rc, ra, li, bl = O.rasterize_fwd(means2d, conics, colors, opac, W, H, 16, offs, flat, backgrounds=bg, return_borderline=True)
    assert rc.dtype == np.float64
    t = lambda a: torch.tensor(a[0], dtype=torch.float64, requires_grad=True)
    m_t, c_t, col_t, o_t = t(means2d), t(conics), t(colors), t(opac)
    d_rgb, d_a, d_last = dense_composite(m_t, c_t, col_t, o_t, torch.tensor(flat.astype(np.int64)), W, H, torch.tensor(bg[0], dtype=torch.float64))
    ok = bl[0] == 0
    assert ok.mean() > 0.98
    # (the dense formulation uses double thresholds 1/255 and 1e-4, the oracle their fp32 values: borderline pixels excluded)
    assert_close(rc[0][ok], d_rgb.detach().numpy()[ok], 1e-7, 1e-9, "colors f64")
    assert_close(ra[0][ok], d_a.detach().numpy()[ok], 1e-7, 1e-9, "alphas f64")
    assert np.array_equal(li[0][ok], d_last.numpy()[ok])
    rs = np.random.RandomState(1)
    v_rc = rs.randn(1, H, W, 3) * ok[None, ..., None]
    v_ra = rs.randn(1, H, W, 1) * ok[None, ..., None]
    loss = (d_rgb * torch.tensor(v_rc[0])).sum() + (d_a * torch.tensor(v_ra[0])).sum()
    g_m, g_c, g_col, g_o = torch.autograd.grad(loss, (m_t, c_t, col_t, o_t))
    with O.precision(64):
        v_m, v_c, v_col, v_o, _ = O.rasterize_bwd(means2d, conics, colors, opac, W, H, 16, offs, flat, ra, li, v_rc, v_ra, backgrounds=bg)
    for name, got, ref in (("v_means2d", v_m[0], g_m), ("v_conics", v_c[0], g_c), ("v_colors", v_col[0], g_col), ("v_opacities", v_o[0], g_o)):
        assert rel_l2(got, ref.numpy()) < 1e-7, (name, rel_l2(got, ref.numpy()))


def test_float64_projection_and_sh_match_the_fp32_oracle():
    """Projection and SH in the float64 build agree with the (reference-pinned) fp32 oracle to fp32 rounding."""
    from util import garden, garden_sh

    fx = garden(1500, scale_mult=3.0)
    W, H = fx["width"], fx["height"]
    args = (fx["means"], None, fx["quats"], fx["scales"], fx["viewmats"][:2], fx["Ks"][:2], W, H)
    r32 = O.projection_fwd(*args)
    with O.precision(64):
        r64 = O.projection_fwd(*args)
    vis = (r32[0] > 0) & (r64[0] > 0)
    assert (r32[0] == r64[0]).mean() > 0.999
    for a, b, name in ((r32[1], r64[1], "means2d"), (r32[2], r64[2], "depths"), (r32[3], r64[3], "conics")):
        assert_close(a[vis], b[vis], 2e-4, 1e-5, name, max_bad_frac=1e-3)
    sh = garden_sh(fx["rgb"])
    dirs = fx["means"][None] - np.linalg.inv(fx["viewmats"][:1])[:, None, :3, 3]
    c32 = O.sh_fwd(3, dirs, sh[None])
    with O.precision(64):
        c64 = O.sh_fwd(3, dirs, sh[None])
    assert_close(c32, c64, 1e-5, 1e-6, "sh")
