"""FULL-SIZE parity of the HIP path against the oracle at BASELINE configs 1 and 2.

config 1  the reference's own CPU-runnable case: garden crop, load_test_data(scene_grid=1) -> 111,785 gaussians, camera 0,
          648 x 420, RGB colours.  Visible splats V = 71,195 and intersections I = 586,348 are the values SURVEY 8(d) records.
config 2  the headline workload: load_test_data(scene_grid=3) -> 1,006,065 gaussians, SH degree 3, 1920 x 1080, camera 0.
          V = 292,931, I ~ 4.0 M (the CPU oracle's own projection gives 3,997,878; a handful of radii differ by one pixel
          between expf / sqrtf on the host and the GPU's, which the reference's own test allows, tests/test_basic.py:246).

For both (the recipes of the reference's tests/test_basic.py:442-472 for binning and tests/test_rasterization.py:17-89 end
to end), on the SAME inputs:
  * radii against the oracle's projection: equal on >= 99.9 % of the splats, never off by more than one pixel;
  * binning BIT-EXACT: the GPU's means2d / radii / depths fed to the oracle's isect_tiles + isect_offset_encode give
    array-equal tiles_per_gauss, all I isect_ids, flatten_ids and offsets;
  * image and alpha within 1e-4 of the oracle's compositing of the GPU's own projected splats (pixels whose threshold
    decisions are within rounding of flipping are flagged by the oracle and excluded; they are < 0.5 % of the image);
  * the five parameter gradients of sum(render * v) within 1e-4 (relative L2) of the FLOAT64 build of the oracle, chained by
    hand: compositing -> SH (config 2) -> projection."""
import math

import numpy as np
import pytest
import torch

from util import N, assert_close, rel_l2

pytestmark = pytest.mark.gpu

from oracle import gs_oracle as O  # noqa: E402


def _f64(a):
    return None if a is None else np.ascontiguousarray(a, np.float64)


def _run_case(w, sh_degree, expect_V=None, expect_I_oracle=None):
    """``w``: workload dict with C >= 1 cameras (viewmats [C,4,4], Ks [C,3,3]).  Returns (V, I, meta, gradients)."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd import _wrapper as ops

    W, H = w["width"], w["height"]
    names = ("means", "quats", "scales", "opacities", "colors")
    P = {k: w[k].clone().requires_grad_(True) for k in names}
    rc, ra, meta = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], w["viewmats"], w["Ks"], W, H,
                                 sh_degree=sh_degree, packed=False)
    host = {k: N(w[k]) for k in names + ("viewmats", "Ks")}
    radii, means2d, depths, conics = (N(meta[k]) for k in ("radii", "means2d", "depths", "conics"))
    vis = radii > 0
    V, I = int(vis.sum()), int(meta["flatten_ids"].numel())
    print(f"[full size] N = {w['means'].shape[0]}  V = {V}  I = {I}")
    C = w["viewmats"].shape[0]
    assert expect_V is None or V == expect_V

    # ---- projection: radii against the oracle's own projection of the same inputs
    o_radii, o_m2, o_dp, o_cn, _ = O.projection_fwd(host["means"], None, host["quats"], host["scales"], host["viewmats"], host["Ks"], W, H)
    same = (o_radii == radii)
    assert same.mean() >= 0.999 and int(np.abs(o_radii - radii).max()) <= 1, (same.mean(), np.abs(o_radii - radii).max())
    assert np.array_equal(o_radii > 0, vis)  # the same splats are visible
    assert_close(means2d[vis], o_m2[vis], 1e-4, 1e-3, "means2d")
    assert_close(depths[vis], o_dp[vis], 1e-5, 1e-6, "depths")
    assert_close(conics[vis], o_cn[vis], 2e-4, 1e-6, "conics", max_bad_frac=1e-4)
    # the oracle's own count (its radii) is within the +-1 pixel radius differences of ours
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    assert expect_I_oracle is None or abs(I - expect_I_oracle) <= 1e-5 * expect_I_oracle, (I, expect_I_oracle)

    # ---- binning, bit-exact on the GPU's own projected values (culled entries are uninitialised: blank them)
    m2z, dpz = np.where(vis[..., None], means2d, 0).astype(np.float32), np.where(vis, depths, 0).astype(np.float32)
    o_tpg, o_ids, o_flat = O.isect_tiles(m2z, radii, dpz, 16, tw, th)
    o_offs = O.isect_offset_encode(o_ids, C, tw, th)
    assert o_ids.size == I
    assert np.array_equal(N(meta["tiles_per_gauss"]), o_tpg)
    assert np.array_equal(N(meta["isect_ids"]), o_ids)
    assert np.array_equal(N(meta["flatten_ids"]), o_flat)
    assert np.array_equal(N(meta["isect_offsets"]), o_offs)

    # ---- compositing forward against the oracle on the GPU's own projected splats
    opac = N(meta["opacities"])
    if sh_degree is None:
        cols = np.ascontiguousarray(np.broadcast_to(host["colors"][None], (C,) + host["colors"].shape))
    else:
        with torch.no_grad():
            cols = N(ops.spherical_harmonics_view(sh_degree, w["means"], w["viewmats"], w["colors"], meta["radii"]))
        cols = np.where(vis[..., None], cols, 0).astype(np.float32)
    cnz = np.where(vis[..., None], conics, 0).astype(np.float32)
    o_rc, o_ra, o_li, bl32 = O.rasterize_fwd(m2z, cnz, cols, opac, W, H, 16, o_offs, o_flat, return_borderline=True)
    ok32 = bl32 == 0
    assert ok32.mean() > 0.995, ok32.mean()
    assert_close(N(rc)[ok32], o_rc[ok32], 1e-4, 1e-5, "render vs oracle", max_bad_frac=2e-5)
    assert_close(N(ra)[ok32], o_ra[ok32], 1e-4, 1e-5, "alpha vs oracle", max_bad_frac=2e-5)

    # ---- float64 ground truth of the whole chain on the same discrete structure (lists, visibility)
    with O.precision(64):
        h64 = {k: _f64(v) for k, v in host.items()}
        _, d_m2, d_dp, d_cn, _ = O.projection_fwd(h64["means"], None, h64["quats"], h64["scales"], h64["viewmats"], h64["Ks"], W, H)
        d_opac = np.ascontiguousarray(np.broadcast_to(h64["opacities"][None], (C,) + h64["opacities"].shape))
        if sh_degree is None:
            d_cols = _f64(cols)
            dirs = shs = None
        else:
            c2w = np.linalg.inv(h64["viewmats"])
            dirs = h64["means"][None] - c2w[:, None, :3, 3]
            shs = h64["colors"][None]  # shared by the cameras: the oracle takes one coefficient row per element -> camera by camera
            sh_raw = np.concatenate([O.sh_fwd(sh_degree, dirs[c:c + 1], shs, vis[c:c + 1]) for c in range(C)], 0)
            d_cols = np.maximum(sh_raw + 0.5, 0.0)
        d_rc, d_ra, d_li, bl64 = O.rasterize_fwd(d_m2, d_cn, d_cols, d_opac, W, H, 16, o_offs, o_flat, return_borderline=True)
    ok = ok32 & (bl64 == 0)
    assert ok.mean() > 0.99, ok.mean()
    # the float64 chain projects in float64 too: its means2d / conics differ from the fp32 ones in the 7th digit, so a few
    # more threshold decisions flip than either build flags on its OWN values; those pixels are excluded as well
    same = o_li == d_li
    assert same[ok].mean() > 0.9995, same[ok].mean()
    ok &= same
    assert_close(N(rc)[ok], d_rc[ok], 1e-4, 1e-5, "render vs f64", max_bad_frac=2e-5)
    # the GPU's own exp (v_exp_f32) flips a handful of threshold decisions that neither oracle build flags (the 2e-5 of the
    # pixels the line above tolerates): a flipped decision changes which splats a pixel feeds gradient to, so those pixels
    # get no upstream gradient either
    agree = (np.abs(N(rc).astype(np.float64) - d_rc) <= 1e-4 * np.abs(d_rc) + 1e-5).all(-1)
    assert (ok & ~agree).mean() <= 2e-5, (ok & ~agree).mean()
    print(f"[full size] pixels without upstream gradient: {(~ok).sum()} flagged borderline / oracle builds differ, "
          f"{(ok & ~agree).sum()} more where the GPU's decision differs")
    ok &= agree

    rs = np.random.RandomState(11)
    v_rc = (rs.randn(*d_rc.shape) * ok[..., None]).astype(np.float32)
    (rc * torch.as_tensor(v_rc, device=rc.device)).sum().backward()

    with O.precision(64):
        v_m2, v_cn, v_col, v_op, _ = O.rasterize_bwd(d_m2, d_cn, d_cols, d_opac, W, H, 16, o_offs, o_flat, d_ra, d_li,
                                                     _f64(v_rc), np.zeros_like(d_ra))
        g_means, _, g_quats, g_scales, _ = O.projection_bwd(h64["means"], None, h64["quats"], h64["scales"], h64["viewmats"], h64["Ks"],
                                                            W, H, 0.3, "pinhole", radii, d_cn, None, v_m2, np.zeros_like(d_dp), v_cn,
                                                            None, need_viewmats=False)
        if sh_degree is None:
            g_colors = v_col.sum(0)
        else:
            g_colors = 0.0
            for c in range(C):
                v_coeffs, v_dirs = O.sh_bwd(sh_degree, dirs[c:c + 1], shs, (v_col * ((sh_raw + 0.5) > 0))[c:c + 1], vis[c:c + 1])
                g_colors = g_colors + v_coeffs[0]
                g_means = g_means + v_dirs[0]
    expect = dict(means=g_means, quats=g_quats, scales=g_scales, opacities=v_op.sum(0), colors=g_colors)
    errs = {}
    for k, ref in expect.items():
        got = N(P[k].grad)
        assert got.shape == ref.shape, k
        errs[k] = rel_l2(got, ref)
        print(f"[full size] d/d {k:10s} rel L2 vs float64 oracle: {errs[k]:.2e}")
    for k, e in errs.items():
        assert e <= 1e-4, (k, e)
    return V, I, meta, {k: P[k].grad for k in names}


def test_config1_full_size_vs_oracle():
    """BASELINE config 1: V = 71,195 and I = 586,348 are the values recorded in SURVEY.md 8(d)."""
    from gscodec_studio_amd._helper import load_test_data

    means, quats, scales, opac, rgb, viewmats, Ks, W, H = load_test_data(device="cuda:0", scene_grid=1)
    assert means.shape[0] == 111785 and (W, H) == (648, 420)
    w = dict(means=means, quats=quats, scales=scales, opacities=opac, colors=rgb, viewmats=viewmats[:1].contiguous(),
             Ks=Ks[:1].contiguous(), width=W, height=H)
    V, I, _, _ = _run_case(w, None, expect_V=71195, expect_I_oracle=586348)
    assert I == 586348


def test_config2_full_size_vs_oracle():
    """BASELINE config 2 (the bench workload): 1,006,065 gaussians, SH degree 3, 1080p."""
    from gscodec_studio_amd._helper import sh_workload

    w = sh_workload(scene_grid=3, device="cuda:0")
    assert w["N"] == 1006065
    w["colors"] = w["sh"]
    V, I, _, _ = _run_case(w, 3, expect_V=292931, expect_I_oracle=3997878)
    assert I == 3997870  # (the GPU's own projection; recorded on MI355X, BENCH_r02.json)


def test_config4_eight_cameras_full_size_vs_oracle():
    """BASELINE config 4's workload on ONE GPU: an 8-camera batch over 1,006,065 gaussians (SH degree 3, 1080p; the three
    fixture cameras and their rolled copies -- eight distinct views with 2x different visible counts).  Unpacked: the full
    recipe above for all 8 cameras at once (binning bit-exact for every camera, image 1e-4, the five parameter gradients --
    sums over the 8 cameras -- within 1e-4 of the float64 oracle chain).  Packed: the same batch through the COO pipeline:
    its own binning bit-exact against the oracle on ITS projection outputs, the visible set equal to the unpacked one up to
    <= 1e-5 of the pairs (the two projection kernels contract their arithmetic differently: a radius at the clip threshold may
    fall on either side), the same image and gradients up to those pairs."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd._helper import sh_workload

    w = sh_workload(scene_grid=3, device="cuda:0", n_cameras=8)
    assert w["N"] == 1006065 and w["viewmats"].shape[0] == 8
    w["colors"] = w["sh"]
    V, I, meta, grads = _run_case(w, 3)
    per_cam = (meta["radii"] > 0).sum(1).tolist()
    print(f"[config 4] visible per camera {per_cam}, I = {I}")
    assert min(per_cam) > 100_000 and len(set(per_cam)) == 8

    rs = np.random.RandomState(11)  # (the same cotangent stream as _run_case; the borderline mask differs: use a fresh dense one)
    v = torch.as_tensor(rs.randn(8, w["height"], w["width"], 3).astype(np.float32), device="cuda:0")
    names = ("means", "quats", "scales", "opacities", "colors")
    out = {}
    for packed in (False, True):
        P = {k: w[k].clone().requires_grad_(True) for k in names}
        rc, ra, m = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], w["viewmats"], w["Ks"],
                                  w["width"], w["height"], sh_degree=3, packed=packed)
        (rc * v).sum().backward()
        out[packed] = (rc.detach(), ra.detach(), m, {k: P[k].grad for k in names})
    (rc_u, ra_u, m_u, g_u), (rc_p, ra_p, m_p, g_p) = out[False], out[True]
    # packed: the COO list holds the visible pairs in row-major order.  The packed projection is a different kernel from the
    # row-writing one: the compiler contracts the same expressions differently, and a pair sitting exactly on a cull
    # threshold can land on the other side (2 of 8 M here) -- the same +-1 pixel class the reference's own test allows
    # (tests/test_basic.py:246).  Binning is checked BIT-EXACT against the oracle on the packed kernel's own values.
    vis = m_u["radii"] > 0
    nnz = m_p["camera_ids"].numel()
    pair_u = torch.nonzero(vis.reshape(-1)).reshape(-1)
    pair_p = m_p["camera_ids"] * w["N"] + m_p["gaussian_ids"]
    assert bool((pair_p[1:] > pair_p[:-1]).all())
    n_diff = nnz + V - 2 * int(torch.isin(pair_p, pair_u).sum())
    print(f"[config 4] packed nnz = {nnz}, unpacked V = {V}, pairs in one list only: {n_diff}")
    assert n_diff <= 1e-5 * V
    tw, th = math.ceil(w["width"] / 16), math.ceil(w["height"] / 16)
    o_tpg, o_ids, o_flat = O.isect_tiles(N(m_p["means2d"]), N(m_p["radii"]), N(m_p["depths"]), 16, tw, th, n_cameras=8,
                                         camera_ids=N(m_p["camera_ids"]))
    assert np.array_equal(N(m_p["tiles_per_gauss"]), o_tpg) and np.array_equal(N(m_p["isect_ids"]), o_ids)
    assert np.array_equal(N(m_p["flatten_ids"]), o_flat)
    assert np.array_equal(N(m_p["isect_offsets"]), O.isect_offset_encode(o_ids, 8, tw, th))
    bad = ((rc_p - rc_u).abs() > 1e-4).any(-1) | ((ra_p - ra_u).abs() > 1e-4).any(-1)
    assert float(bad.float().mean()) <= 1e-5, float(bad.float().mean())
    for k in names:
        e = float((g_p[k] - g_u[k]).norm() / g_u[k].norm())
        print(f"[config 4] packed vs unpacked d/d {k}: {e:.2e}")
        assert e <= 1e-4, (k, e)
