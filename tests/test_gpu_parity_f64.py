"""GPU parity against a FLOAT64 ground truth (oracle/gs_oracle.c compiled with float -> double, pinned on the CPU against
the dense float64 autograd formulation in tests/test_oracle_raster.py).

The fp32 oracle and the HIP kernels are two fp32 evaluations of the same algorithm (expf vs v_exp_f32, double vs
float per-splat sums, different summation orders).  Comparing them with each other cannot tell which one carries an
error; comparing BOTH with float64 can.  For every gradient this file asserts

  * rel_l2(HIP, f64) <= 1e-4                      -- the north star's "grads within 1e-4 rel fp32"
    (measured on MI355X: 7e-6 ... 3e-5 for the compositing gradients, where the fp32 ORACLE is at 4e-5 ... 2e-4: the
    reference's back-to-front T <- T / (1 - alpha) recurrence over the whole list loses digits that the segmented
    backward, restarting from forward checkpoints every 256 entries, keeps);
  * rel_l2(HIP, f64) <= FACTOR * rel_l2(oracle_f32, f64) + 2e-6   -- the HIP path is as accurate as fp32 gets
    (FACTOR 4: the oracle accumulates its per-splat sums in double, the kernels in fp32 registers and float atomics);
  * max |HIP - f64| <= 1e-4 max |f64|;
  * entrywise 1e-4 relative is reached on at least as many entries (-1 %) as the fp32 oracle reaches it on, and for the
    position gradient |HIP - f64| <= 1e-4 |f64| + 1e-4 * cond with cond = sum_pixels |per-pixel term| (the absgrad of the
    float64 run): an entry that is the small difference of large per-pixel terms cannot be relatively accurate in ANY
    fp32 evaluation, the fp32 oracle included (measured: the oracle reaches 1e-4 on ~90 % of the v_means2d entries).

Threshold decisions (alpha >= 1/255, T(1 - alpha) <= 1e-4, sigma < 0) within a few ulp of flipping differ between exp
implementations; such pixels are flagged by the oracle, get zero upstream gradient here, and their FORWARD error is
bounded separately (test_borderline_pixels_are_bounded)."""
import math

import numpy as np
import pytest
import torch

from util import N, T, assert_close, garden, garden_sh, rel_l2

pytestmark = pytest.mark.gpu

from oracle import gs_oracle as O  # noqa: E402

FACTOR = 4.0


def _case(n=4000, scale_mult=6.0, cams=2, channels=3, seed=0, opac_boost=True):
    fx = garden(n, scale_mult=scale_mult)
    W, H = fx["width"], fx["height"]
    radii, means2d, depths, conics, _ = O.projection_fwd(fx["means"], None, fx["quats"], fx["scales"], fx["viewmats"][:cams],
                                                         fx["Ks"][:cams], W, H)
    rs = np.random.RandomState(seed)
    opac = np.broadcast_to(fx["opacities"][None], (cams, n)).copy()
    if opac_boost:
        opac = np.clip(opac * 3.0, 0, 1).astype(np.float32)
    colors = rs.rand(cams, n, channels).astype(np.float32)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = O.isect_tiles(means2d, radii, depths, 16, tw, th)
    offs = O.isect_offset_encode(ids, cams, tw, th)
    return dict(means2d=means2d, conics=conics, colors=colors, opacities=opac, W=W, H=H, offs=offs, flat=flat, C=cams)


def _check(name, hip, f32, f64, cond=None):
    e_hip, e_orc = rel_l2(hip, f64), rel_l2(f32, f64)
    print(f"[f64 ground truth] {name:12s} rel_l2: HIP {e_hip:.2e}  fp32 oracle {e_orc:.2e}")
    assert e_hip <= 1e-4, (name, e_hip)
    assert e_hip <= FACTOR * e_orc + 2e-6, (name, e_hip, e_orc)
    f64 = np.asarray(f64, np.float64)
    hip, f32 = np.asarray(hip, np.float64), np.asarray(f32, np.float64)
    floor = 1e-7 * np.abs(f64).max()
    # max norm: no entry is off by more than 1e-4 of the largest entry -- or, where the fp32 ORACLE itself is further off than
    # that (32 channels: 1.35e-4 for v_opacities, the kernels 1.13e-4), by more than the oracle is
    # (round-5 advisor: the adaptive bound is CAPPED at 2e-4, so that a regression of the kernels cannot hide behind a bad day of the fp32
    # oracle, and the observed error is printed per case so that drift stays visible when the bound moves)
    bound = min(2e-4, max(1e-4, np.abs(f32 - f64).max() / np.abs(f64).max()))
    e_max = np.abs(hip - f64).max() / np.abs(f64).max()
    print(f"                   {name:12s} max norm: HIP {e_max:.2e}  bound {bound:.2e}")
    assert e_max <= bound, (name, "max norm", e_max, bound)
    # entrywise 1e-4 relative is reached on (at least) as many entries as the fp32 oracle reaches it on: the entries that
    # miss it are small differences of large per-pixel terms, which no fp32 evaluation resolves (see `cond` below)
    reach_o = (np.abs(f32 - f64) <= 1e-4 * np.abs(f64) + floor).mean()
    reach_h = (np.abs(hip - f64) <= 1e-4 * np.abs(f64) + floor).mean()
    print(f"                   {name:12s} entries within 1e-4 relative: HIP {reach_h*100:.2f}%  fp32 oracle {reach_o*100:.2f}%")
    assert reach_h >= reach_o - 0.01, (name, reach_h, reach_o)
    if cond is not None:
        bad = np.abs(hip - f64) > 1e-4 * np.abs(f64) + 1e-4 * np.asarray(cond, np.float64) + floor
        assert bad.mean() <= 1e-5, (name, "conditioned bound", bad.mean())


# (round 5: 9 and 32 channels -- the wide instances, 32 as two launches over halves -- without absgrad: with it the 32-channel
# backward is the generic kernel, which the absgrad=True cases of tests/test_gpu_ops.py cover)
@pytest.mark.parametrize("channels,absgrad", [(3, True), (1, True), (9, False), (9, True), (16, False), (32, False)])
def test_compositing_gradients_vs_float64(channels, absgrad):
    from gscodec_studio_amd import _wrapper as ops

    c = _case(channels=channels, n=4000 if channels <= 9 else 2500)
    rs = np.random.RandomState(5)
    bg = rs.rand(c["C"], channels).astype(np.float32)
    geo = (c["means2d"], c["conics"], c["colors"], c["opacities"], c["W"], c["H"], 16, c["offs"], c["flat"])
    o_rc, o_ra, o_li, bl32 = O.rasterize_fwd(*geo, backgrounds=bg, return_borderline=True)
    with O.precision(64):
        d_rc, d_ra, d_li, bl64 = O.rasterize_fwd(*geo, backgrounds=bg, return_borderline=True)
    ok = (bl32 == 0) & (bl64 == 0)
    assert ok.mean() > 0.995
    assert np.array_equal(o_li[ok], d_li[ok])  # same decisions on every pixel that is not flagged
    m2, cn, col, op = T(c["means2d"], True), T(c["conics"], True), T(c["colors"], True), T(c["opacities"], True)
    rc, ra = ops.rasterize_to_pixels(m2, cn, col, op, c["W"], c["H"], 16, T(c["offs"]), T(c["flat"]), backgrounds=T(bg), absgrad=absgrad)
    # forward against float64: 1e-4 relative (+ 1e-6 absolute: colours are O(1))
    assert_close(N(rc)[ok], d_rc[ok], 1e-4, 1e-6, "render vs f64", max_bad_frac=1e-5)
    assert_close(N(ra)[ok], d_ra[ok], 1e-4, 1e-6, "alpha vs f64", max_bad_frac=1e-5)

    v_rc = rs.randn(*o_rc.shape).astype(np.float32) * ok[..., None]
    v_ra = rs.randn(*o_ra.shape).astype(np.float32) * ok[..., None]
    loss = (rc * T(v_rc)).sum() + (ra * T(v_ra)).sum()
    g = torch.autograd.grad(loss, (m2, cn, col, op))
    o = O.rasterize_bwd(*geo, o_ra, o_li, v_rc, v_ra, backgrounds=bg)
    with O.precision(64):
        d = O.rasterize_bwd(*geo, d_ra, d_li, v_rc, v_ra, backgrounds=bg, absgrad=True)
    # conditioning scale of the position gradient: sum_pixels |per-pixel term| (the absgrad of the float64 run)
    _check("v_means2d", N(g[0]), o[0], d[0], cond=d[4])
    _check("v_conics", N(g[1]), o[1], d[1])
    _check("v_colors", N(g[2]), o[2], d[2])
    _check("v_opacities", N(g[3]), o[3], d[3])


def test_projection_gradients_vs_float64():
    """fully_fused_projection backward (quat + scale route, pinhole) against the float64 build of the oracle."""
    from gscodec_studio_amd import fully_fused_projection

    fx = garden(6000, scale_mult=3.0)
    W, H = fx["width"], fx["height"]
    cams = 3
    vm, Ks = fx["viewmats"][:cams], fx["Ks"][:cams]
    m, q, s = T(fx["means"], True), T(fx["quats"], True), T(fx["scales"], True)
    radii, means2d, depths, conics, _ = fully_fused_projection(m, None, q, s, T(vm), T(Ks), W, H, packed=False)
    r32 = O.projection_fwd(fx["means"], None, fx["quats"], fx["scales"], vm, Ks, W, H)
    with O.precision(64):
        r64 = O.projection_fwd(fx["means"], None, fx["quats"], fx["scales"], vm, Ks, W, H)
    vis = (N(radii) > 0) & (r32[0] > 0) & (r64[0] > 0)
    assert vis.sum() > 0.99 * (r64[0] > 0).sum()
    assert_close(N(means2d)[vis], r64[1][vis], 1e-4, 1e-3, "means2d vs f64")  # pixels: 1e-3 px absolute floor
    assert_close(N(depths)[vis], r64[2][vis], 1e-5, 1e-6, "depths vs f64")
    assert_close(N(conics)[vis], r64[3][vis], 1e-4, 1e-7, "conics vs f64", max_bad_frac=2e-4)
    rs = np.random.RandomState(3)
    v2 = (rs.randn(*r32[1].shape) * vis[..., None]).astype(np.float32)
    vd = (rs.randn(*r32[2].shape) * vis).astype(np.float32)
    vc = (rs.randn(*r32[3].shape) * vis[..., None]).astype(np.float32)
    loss = (means2d * T(v2)).sum() + (depths * T(vd)).sum() + (conics * T(vc)).sum()
    g = torch.autograd.grad(loss, (m, q, s))
    # both oracles differentiate at THEIR forward state, restricted to the same visible set
    rad = (vis * np.maximum(r32[0], 1)).astype(np.int32)
    o = O.projection_bwd(fx["means"], None, fx["quats"], fx["scales"], vm, Ks, W, H, 0.3, "pinhole", rad, r32[3], None, v2, vd, vc,
                         None, need_viewmats=False)
    with O.precision(64):
        d = O.projection_bwd(fx["means"], None, fx["quats"], fx["scales"], vm, Ks, W, H, 0.3, "pinhole", rad, r64[3], None, v2, vd,
                             vc, None, need_viewmats=False)
    _check("v_means", N(g[0]), o[0], d[0])
    _check("v_quats", N(g[1]), o[2], d[2])
    _check("v_scales", N(g[2]), o[3], d[3])


def test_sh_gradients_vs_float64():
    from gscodec_studio_amd import spherical_harmonics

    rs = np.random.RandomState(42)
    n, K = 20000, 16
    dirs = rs.randn(n, 3).astype(np.float32)
    coeffs = rs.randn(n, K, 3).astype(np.float32)
    v = rs.randn(n, 3).astype(np.float32)
    d_t, c_t = T(dirs, True), T(coeffs, True)
    out = spherical_harmonics(3, d_t, c_t)
    g_d, g_c = torch.autograd.grad((out * T(v)).sum(), (d_t, c_t))
    o_c, o_d = O.sh_bwd(3, dirs, coeffs, v)
    with O.precision(64):
        f_out = O.sh_fwd(3, dirs, coeffs)
        d_c, d_d = O.sh_bwd(3, dirs, coeffs, v)
    assert_close(N(out), f_out, 1e-4, 1e-5, "sh colours vs f64")
    _check("v_coeffs", N(g_c), o_c, d_c)
    _check("v_dirs", N(g_d), o_d, d_d)


@pytest.mark.parametrize("channels", [3, 1])
def test_borderline_pixels_are_bounded(channels):
    """The pixels the parity tests exclude (a threshold decision within a few ulp of flipping) are not a blind spot: there
    the render may differ from the oracle by ONE list entry entering or leaving the sum, i.e. by at most
    2 x (largest single blending weight of the pixel) x max|colour| in colour and that weight in alpha."""
    from gscodec_studio_amd import _wrapper as ops

    c = _case(channels=channels, n=4000, scale_mult=8.0, seed=3)
    geo = (c["means2d"], c["conics"], c["colors"], c["opacities"], c["W"], c["H"], 16, c["offs"], c["flat"])
    o_rc, o_ra, o_li, bl = O.rasterize_fwd(*geo, return_borderline=True)
    mw = O.rasterize_max_weight(c["means2d"], c["conics"], c["opacities"], c["W"], c["H"], 16, c["offs"], c["flat"])
    rc, ra = ops.rasterize_to_pixels(T(c["means2d"]), T(c["conics"]), T(c["colors"]), T(c["opacities"]), c["W"], c["H"], 16,
                                     T(c["offs"]), T(c["flat"]))
    flagged = bl != 0
    assert flagged.sum() > 0, "the case must contain borderline pixels"
    cmax = float(np.abs(c["colors"]).max())
    err_c = np.abs(N(rc) - o_rc).max(-1)
    err_a = np.abs(N(ra) - o_ra)[..., 0]
    bound_c = 2.0 * mw * cmax * (1 + 1e-3) + 2e-5
    bound_a = mw * (1 + 1e-3) + 2e-5
    assert (err_c[flagged] <= bound_c[flagged]).all(), float((err_c - bound_c)[flagged].max())
    assert (err_a[flagged] <= bound_a[flagged]).all(), float((err_a - bound_a)[flagged].max())
    # and the flips that do happen are rare: the vast majority of flagged pixels still agree to 1e-4
    agree = err_c[flagged] <= 1e-4 * np.abs(o_rc).max(-1)[flagged] + 2e-5
    print(f"[borderline] {flagged.sum()} flagged pixels ({flagged.mean()*100:.3f}%), {100 - agree.mean()*100:.2f}% of them actually differ")
