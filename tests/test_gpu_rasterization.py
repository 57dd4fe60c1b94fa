"""End-to-end GPU parity of rasterization() against the CPU oracle (the reference's
tests/test_rasterization.py:17-89 compares rasterization() with its torch twin the same way),
plus size-independent properties at BASELINE config-2 size."""
import numpy as np
import pytest
import torch

from util import N, T, assert_close, garden, garden_sh, rel_l2

pytestmark = pytest.mark.gpu

from oracle import gs_oracle as O  # noqa: E402


def _inputs(n=3000, scale_mult=5.0, cams=2, sh_degree=3):
    fx = garden(n, scale_mult=scale_mult)
    d = dict(means=fx["means"], quats=fx["quats"], scales=fx["scales"], opacities=fx["opacities"],
             viewmats=fx["viewmats"][:cams], Ks=fx["Ks"][:cams], W=fx["width"], H=fx["height"])
    d["colors"] = garden_sh(fx["rgb"], K=16) if sh_degree is not None else fx["rgb"]
    return d


@pytest.mark.parametrize("sh_degree", [None, 3])
@pytest.mark.parametrize("packed", [False, True])
@pytest.mark.parametrize("render_mode", ["RGB", "RGB+D", "D", "RGB+ED"])
def test_rasterization_vs_oracle(sh_degree, packed, render_mode):
    from gscodec_studio_amd import rasterization

    d = _inputs(sh_degree=sh_degree)
    rc, ra, meta = rasterization(T(d["means"]), T(d["quats"]), T(d["scales"]), T(d["opacities"]), T(d["colors"]),
                                 T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"], sh_degree=sh_degree, packed=packed,
                                 render_mode=render_mode)
    o_rc, o_ra, om = O.rasterization(d["means"], d["quats"], d["scales"], d["opacities"], d["colors"], d["viewmats"],
                                     d["Ks"], d["W"], d["H"], sh_degree=sh_degree)
    C = d["viewmats"].shape[0]
    assert rc.shape[:3] == (C, d["H"], d["W"]) and ra.shape == (C, d["H"], d["W"], 1)
    for k in ("camera_ids", "gaussian_ids", "radii", "means2d", "depths", "conics", "opacities", "tile_width",
              "tile_height", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets", "width", "height",
              "tile_size", "n_cameras"):
        assert k in meta, k
    if not packed:
        assert meta["camera_ids"] is None and meta["gaussian_ids"] is None
        # integer meta: exact where the projection agrees (it does on >99.9% of splats)
        same = N(meta["radii"]) == om["radii"]
        assert same.mean() > 0.999
    else:
        assert meta["camera_ids"].dtype == torch.int64 and meta["gaussian_ids"].dtype == torch.int64
    # depth channel of the oracle (channel = camera-space z of each splat)
    if render_mode != "RGB":
        cols = om["colors"]
        dep = om["depths"][..., None]
        cols = dep if render_mode == "D" else np.concatenate([cols, dep], -1)
        o_rc, o_ra, _ = O.rasterize_fwd(om["means2d"], om["conics"], cols, om["opacities"], d["W"], d["H"], 16,
                                        om["isect_offsets"], om["flatten_ids"])
        if render_mode == "RGB+ED":
            o_rc = np.concatenate([o_rc[..., :-1], o_rc[..., -1:] / np.clip(o_ra, 1e-10, None)], -1)
    assert_close(N(ra), o_ra, 1e-4, 5e-5, "alphas", max_bad_frac=2e-4)
    assert_close(N(rc), o_rc, 1e-4, 5e-5, "colors", max_bad_frac=2e-4)


@pytest.mark.parametrize("channels", [9, 32])
@pytest.mark.parametrize("packed", [False, True])
def test_feature_render_forward_backward_vs_oracle_chain(channels, packed):
    """The reference's spacetime trainer renders a 9-channel feature image every step -- colors = cat(feature_color,
    feature_dir, t * feature_time), examples/simple_trainer_STG.py:531-551 -- and its profile publishes a 32-channel row
    (docs/source/tests/profile.rst:76-93): `rasterization()` with [N, D] post-activation colours through the wide compositing
    kernels (round 5), forward against the oracle's pipeline and the full gradient chain against the hand-chained oracle VJPs,
    packed and unpacked, with a background."""
    from gscodec_studio_amd import rasterization

    d = _inputs(n=2500, cams=2, sh_degree=None)
    rs = np.random.RandomState(3 + channels)
    feats = rs.rand(d["means"].shape[0], channels).astype(np.float32)
    bg = rs.rand(2, channels).astype(np.float32)
    P = {k: T(d[k], True) for k in ("means", "quats", "scales", "opacities")}
    P["colors"] = T(feats, True)
    rc, ra, meta = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], T(d["viewmats"]), T(d["Ks"]),
                                 d["W"], d["H"], packed=packed, backgrounds=T(bg))
    assert rc.shape == (2, d["H"], d["W"], channels)
    o_rc, o_ra, om = O.rasterization(d["means"], d["quats"], d["scales"], d["opacities"], feats, d["viewmats"], d["Ks"], d["W"], d["H"])
    cols = np.ascontiguousarray(np.broadcast_to(feats[None], (2,) + feats.shape))
    o_rc, o_ra, o_li, bl = O.rasterize_fwd(om["means2d"], om["conics"], cols, om["opacities"], d["W"], d["H"], 16, om["isect_offsets"],
                                           om["flatten_ids"], backgrounds=bg, return_borderline=True)
    ok = bl == 0
    assert ok.mean() > 0.99
    assert_close(N(ra)[ok], o_ra[ok], 1e-4, 5e-5, "alphas", max_bad_frac=2e-4)
    assert_close(N(rc)[ok], o_rc[ok], 1e-4, 5e-5, "features", max_bad_frac=2e-4)
    v_rc = rs.randn(*o_rc.shape).astype(np.float32) * ok[..., None]
    v_ra = rs.randn(*o_ra.shape).astype(np.float32) * ok[..., None]
    ((rc * T(v_rc)).sum() + (ra * T(v_ra)).sum()).backward()
    v_m2, v_cn, v_col, v_op, _ = O.rasterize_bwd(om["means2d"], om["conics"], cols, om["opacities"], d["W"], d["H"], 16, om["isect_offsets"],
                                                 om["flatten_ids"], o_ra, o_li, v_rc, v_ra, backgrounds=bg)
    vm, _, vq, vs, _ = O.projection_bwd(d["means"], None, d["quats"], d["scales"], d["viewmats"], d["Ks"], d["W"], d["H"], 0.3, "pinhole",
                                        om["radii"], om["conics"], None, v_m2, np.zeros_like(om["depths"]), v_cn, None, need_viewmats=False)
    for name, ref in (("means", vm), ("quats", vq), ("scales", vs), ("opacities", v_op.sum(0)), ("colors", v_col.sum(0))):
        got = N(P[name].grad)
        assert rel_l2(got, ref) < 2e-3, (name, channels, packed, rel_l2(got, ref))


def test_rasterization_backward_vs_oracle_chain():
    """Full chain gradient (quantizer-free): d(sum of weighted render)/d(params) on the GPU vs the
    oracle stage VJPs chained by hand."""
    from gscodec_studio_amd import rasterization

    d = _inputs(n=2500, cams=2, sh_degree=3)
    P = {k: T(d[k], True) for k in ("means", "quats", "scales", "opacities", "colors")}
    rc, ra, meta = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], T(d["viewmats"]),
                                 T(d["Ks"]), d["W"], d["H"], sh_degree=3, packed=False, absgrad=True)
    meta["means2d"].retain_grad()  # DefaultStrategy contract (reference strategy/default.py:150)
    rs = np.random.RandomState(11)
    o_rc, o_ra, om = O.rasterization(d["means"], d["quats"], d["scales"], d["opacities"], d["colors"], d["viewmats"],
                                     d["Ks"], d["W"], d["H"], sh_degree=3)
    _, _, _, bl = O.rasterize_fwd(om["means2d"], om["conics"], om["colors"], om["opacities"], d["W"], d["H"], 16,
                                  om["isect_offsets"], om["flatten_ids"], return_borderline=True)
    v_rc = rs.randn(*o_rc.shape).astype(np.float32) * (bl == 0)[..., None]
    v_ra = rs.randn(*o_ra.shape).astype(np.float32) * (bl == 0)[..., None]
    ((rc * T(v_rc)).sum() + (ra * T(v_ra)).sum()).backward()
    assert meta["means2d"].grad is not None and meta["means2d"].absgrad is not None

    # oracle chain
    C, Ng = om["radii"].shape
    v_m2, v_cn, v_col, v_op, _ = O.rasterize_bwd(om["means2d"], om["conics"], om["colors"], om["opacities"], d["W"],
                                                 d["H"], 16, om["isect_offsets"], om["flatten_ids"], o_ra,
                                                 om["last_ids"], v_rc, v_ra)
    assert rel_l2(N(meta["means2d"].grad), v_m2) < 5e-4
    c2w = np.linalg.inv(d["viewmats"].astype(np.float64)).astype(np.float32)
    dirs = d["means"][None] - c2w[:, None, :3, 3]
    sh_raw = O.sh_fwd(3, dirs, np.ascontiguousarray(np.broadcast_to(d["colors"][None], (C,) + d["colors"].shape)), om["radii"] > 0)
    v_sh_out = v_col * ((sh_raw + 0.5) > 0)  # clamp_min(x + 0.5, 0)
    v_coeffs, v_dirs = O.sh_bwd(3, dirs, np.ascontiguousarray(np.broadcast_to(d["colors"][None], (C,) + d["colors"].shape)),
                                v_sh_out, om["radii"] > 0)
    vm, _, vq, vs, _ = O.projection_bwd(d["means"], None, d["quats"], d["scales"], d["viewmats"], d["Ks"], d["W"], d["H"],
                                        0.3, "pinhole", om["radii"], om["conics"], None, v_m2,
                                        np.zeros_like(om["depths"]), v_cn, None, need_viewmats=False)
    vm = vm + v_dirs.sum(0)
    for name, ref in (("means", vm), ("quats", vq), ("scales", vs), ("opacities", v_op.sum(0)), ("colors", v_coeffs.sum(0))):
        got = N(P[name].grad)
        assert rel_l2(got, ref) < 2e-3, (name, rel_l2(got, ref))


def test_rasterization_antialiased_fisheye_covars_backgrounds():
    from gscodec_studio_amd import rasterization

    d = _inputs(n=2000, cams=1, sh_degree=None)
    bg = np.array([[0.2, 0.5, 0.9]], np.float32)
    rc, ra, meta = rasterization(T(d["means"]), T(d["quats"]), T(d["scales"]), T(d["opacities"]), T(d["colors"]),
                                 T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"], packed=False, backgrounds=T(bg),
                                 rasterize_mode="antialiased", camera_model="fisheye")
    o_rc, o_ra, om = O.rasterization(d["means"], d["quats"], d["scales"], d["opacities"], d["colors"], d["viewmats"],
                                     d["Ks"], d["W"], d["H"], backgrounds=bg, camera_model="fisheye", antialiased=True)
    assert_close(N(rc), o_rc, 1e-4, 5e-5, "fisheye/antialiased colors", max_bad_frac=5e-4)
    # (meta["opacities"] is a column of the splat rows: like means2d / conics it is only defined where radii > 0)
    vis = (om["radii"] > 0) & (N(meta["radii"]) > 0)
    assert_close(N(meta["opacities"])[vis], om["opacities"][vis], 1e-3, 1e-3, "compensated opacities")
    # covars instead of quats/scales
    from gscodec_studio_amd import quat_scale_to_covar_preci

    cov, _ = quat_scale_to_covar_preci(T(d["quats"]), T(d["scales"]), compute_preci=False)
    rc2, _, _ = rasterization(T(d["means"]), None, None, T(d["opacities"]), T(d["colors"]), T(d["viewmats"]), T(d["Ks"]),
                              d["W"], d["H"], packed=False, covars=cov)
    rc3, _, _ = rasterization(T(d["means"]), T(d["quats"]), T(d["scales"]), T(d["opacities"]), T(d["colors"]),
                              T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"], packed=False)
    assert_close(N(rc2), N(rc3), 1e-3, 1e-4, "covars path", max_bad_frac=1e-3)


def test_packed_equals_unpacked_and_sparse_grad():
    """reference tests/test_basic.py:282-439"""
    from gscodec_studio_amd import fully_fused_projection

    fx = garden(3000, scale_mult=4.0)
    args = (T(fx["viewmats"]), T(fx["Ks"]), fx["width"], fx["height"])
    m, q, s = T(fx["means"], True), T(fx["quats"], True), T(fx["scales"], True)
    radii, means2d, depths, conics, _ = fully_fused_projection(m, None, q, s, *args, packed=False)
    cam, gau, radii_p, means2d_p, depths_p, conics_p, _ = fully_fused_projection(m, None, q, s, *args, packed=True)
    cam_n, gau_n = N(cam), N(gau)
    # rows sorted by (camera, gaussian)
    key = cam_n * 10**7 + gau_n
    assert (np.diff(key) > 0).all()
    dense_r = np.zeros(N(radii).shape, np.int32)
    dense_r[cam_n, gau_n] = N(radii_p)
    sel = (N(radii) > 0) & (dense_r > 0)
    assert np.abs(dense_r - N(radii))[sel].max() <= 1  # the two radius formulas
    assert sel.sum() > 0.98 * (N(radii) > 0).sum()
    both = sel[cam_n, gau_n]
    assert_close(N(means2d_p)[both], N(means2d)[cam_n, gau_n][both], 1e-4, 2e-4, "packed means2d")
    assert_close(N(conics_p)[both], N(conics)[cam_n, gau_n][both], 3e-4, 1e-5, "packed conics")
    v2, vd, vc = torch.randn_like(means2d_p), torch.randn_like(depths_p), torch.randn_like(conics_p)
    loss_p = (means2d_p * v2).sum() + (depths_p * vd).sum() + (conics_p * vc).sum()
    g_p = torch.autograd.grad(loss_p, (m, q, s), retain_graph=True)
    V2, VD, VC = torch.zeros_like(means2d), torch.zeros_like(depths), torch.zeros_like(conics)
    V2[cam, gau], VD[cam, gau], VC[cam, gau] = v2, vd, vc
    keep = (radii > 0)
    loss_u = (means2d * V2 * keep[..., None]).sum() + (depths * VD * keep).sum() + (conics * VC * keep[..., None]).sum()
    g_u = torch.autograd.grad(loss_u, (m, q, s))
    only_p = torch.ones(len(cam_n), dtype=torch.bool, device=cam.device)
    only_p[torch.as_tensor(both, device=cam.device)] = False
    if int(only_p.sum()) == 0:
        for a, b in zip(g_p, g_u):
            assert rel_l2(N(a), N(b)) < 1e-4
    # sparse grads densify to the dense ones
    out = fully_fused_projection(m, None, q, s, *args, packed=True, sparse_grad=True)
    loss_s = (out[3] * v2).sum() + (out[4] * vd).sum() + (out[5] * vc).sum()
    g_s = torch.autograd.grad(loss_s, (m, q, s))
    for a, b in zip(g_s, g_p):
        assert a.is_sparse
        assert rel_l2(N(a.to_dense()), N(b)) < 1e-5


def test_full_size_properties():
    """BASELINE config 2 size (1,006,065 gaussians, SH deg 3, 1080p): size-independent checks."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd._helper import sh_workload

    w = sh_workload(scene_grid=3, device="cuda:0")
    P = {k: w[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")}
    rc, ra, meta = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["sh"], w["viewmats"], w["Ks"],
                                 w["width"], w["height"], sh_degree=3, packed=False)
    ids, flat, offs = meta["isect_ids"], meta["flatten_ids"], meta["isect_offsets"]
    n_isects = ids.numel()
    assert w["N"] == 1006065
    # sortedness of the 64-bit keys; every flatten id refers to a visible splat; counts add up
    assert bool((ids[1:] >= ids[:-1]).all())
    assert int(meta["tiles_per_gauss"].sum()) == n_isects
    assert bool((meta["radii"].flatten()[flat.long()] > 0).all())
    # offsets: monotone, consistent with the tile id stored in the keys
    o = offs.flatten().long()
    assert bool((o[1:] >= o[:-1]).all()) and int(o[0]) == 0
    tile_of = ((ids >> 32) & ((1 << 13) - 1)).long()
    counts = torch.bincount(tile_of, minlength=o.numel())
    assert torch.equal(torch.cat([o[1:], o.new_tensor([n_isects])]) - o, counts)
    # image sanity: alpha in [0,1], colours finite and >= 0 (clamp_min(sh + 0.5, 0) inputs)
    assert float(ra.detach().min()) >= 0 and float(ra.detach().max()) <= 1.0 and bool(torch.isfinite(rc).all()) and float(rc.detach().min()) >= 0
    # linearity of compositing in the colours: render(2*sh0-shifted colours) -- use the gradient instead:
    (rc.sum()).backward()
    for k, p in P.items():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
    # d(sum render)/d(sh0 coefficient) = SH_C0 * sum_pixels(weight) >= 0 for unclamped colours
    vis = (meta["radii"][0] > 0)
    assert float(P["sh"].grad[~vis].abs().max()) == 0.0  # culled splats get exactly zero gradient


def test_camera_centers_and_pose_gradients():
    """Closed-form camera centres == inverse(viewmats)[:, :3, 3] (also for non-rigid affine matrices),
    and the unfused autograd route is taken (and works) when the poses require grad."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd.rendering import _camera_centers

    d = _inputs(n=1500, cams=2, sh_degree=3)
    V = T(d["viewmats"])
    V2 = V.clone()
    V2[:, :3, :3] = V2[:, :3, :3] * 1.3 + 0.05 * torch.randn(2, 3, 3, device=V.device)
    for M in (V, V2):
        ref = torch.linalg.inv(M.double().cpu())[:, :3, 3].float()
        assert_close(N(_camera_centers(M)), ref.numpy(), 1e-5, 1e-5, "kernel path")
        assert_close(N(_camera_centers(M.clone().requires_grad_(True))), ref.numpy(), 1e-5, 1e-5, "autograd path")
    Vg = V.clone().requires_grad_(True)
    rc, ra, _ = rasterization(T(d["means"]), T(d["quats"]), T(d["scales"]), T(d["opacities"]), T(d["colors"]), Vg,
                              T(d["Ks"]), d["W"], d["H"], sh_degree=3, packed=False)
    rc2, _, _ = rasterization(T(d["means"]), T(d["quats"]), T(d["scales"]), T(d["opacities"]), T(d["colors"]), V,
                              T(d["Ks"]), d["W"], d["H"], sh_degree=3, packed=False)
    assert_close(N(rc), N(rc2), 1e-5, 1e-6, "fused vs unfused SH route", max_bad_frac=1e-4)
    rc.sum().backward()
    assert Vg.grad is not None and bool(torch.isfinite(Vg.grad).all()) and float(Vg.grad.abs().sum()) > 0


def test_solo_waves_match_cooperative_tiles():
    """Long lists are composited by four independent waves (each walks the whole list and culls against its own quadrant)
    instead of the cooperative workgroup (tuning "raster_solo_min" = list length from which that happens, 2048 by default).  Same
    records in the same order with the same arithmetic: images, alphas, last ids (through the gradients) and the
    checkpoints the backward restarts from must be IDENTICAL, with and without backgrounds, for a mix of both kinds."""
    from gscodec_studio_amd import rasterization

    d = _inputs(n=4000, cams=2, sh_degree=None, scale_mult=12.0)

    def run(bg):
        ps = [T(d[k]).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")]
        rc, ra, meta = rasterization(*ps, T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"], packed=False, backgrounds=bg)
        w = torch.linspace(0.5, 1.5, rc.numel(), device="cuda").reshape(rc.shape)
        ((rc * w).sum() + 0.3 * ra.sum()).backward()
        return N(rc), N(ra), [N(p.grad) for p in ps], meta

    from util import tuned

    for bg in (None, torch.tensor([[0.2, 0.4, 0.6], [0.1, 0.1, 0.9]], device="cuda")):
        with tuned("no_solo"):  # every tile cooperative
            rc0, ra0, g0, meta = run(bg)
        offs = N(meta["isect_offsets"]).reshape(-1)
        lens = np.diff(np.concatenate([offs, [meta["flatten_ids"].numel()]]))
        for thr in (1, int(np.median(lens[lens > 0]))):  # every tile solo / about half of them
            with tuned({"raster_solo_min": thr}):
                rc1, ra1, g1, _ = run(bg)
            assert np.array_equal(rc1, rc0) and np.array_equal(ra1, ra0), thr
            for a, b, name in zip(g1, g0, ("means", "quats", "scales", "opacities", "colors")):
                assert rel_l2(a, b) < 3e-4, (thr, name, rel_l2(a, b))  # (float atomics: the summation order varies run to run)


def test_full_size_linearity_determinism_and_adjoint():
    """BASELINE config 2 size, post-activation colours [N,3]: the forward is bit-reproducible (no atomics), exactly
    linear in the colours (same weights), and the backward is its adjoint: <W, render(c)> == <v_colors(W), c>."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd._helper import sh_workload

    w = sh_workload(scene_grid=3, device="cuda:0")
    g = torch.Generator(device="cuda").manual_seed(1)
    c1 = torch.rand(w["N"], 3, device="cuda", generator=g)
    c2 = torch.rand(w["N"], 3, device="cuda", generator=g)
    args = (w["means"], w["quats"], w["scales"], w["opacities"])
    kw = dict(viewmats=w["viewmats"], Ks=w["Ks"], width=w["width"], height=w["height"], packed=False)
    r1, a1, _ = rasterization(*args, c1, **kw)
    r1b, a1b, _ = rasterization(*args, c1, **kw)
    assert torch.equal(r1, r1b) and torch.equal(a1, a1b)  # deterministic forward
    r2, _, _ = rasterization(*args, c2, **kw)
    r12, _, _ = rasterization(*args, c1 + 2.0 * c2, **kw)
    lin = r1 + 2.0 * r2
    assert float((r12 - lin).abs().max()) <= 1e-4 * float(lin.abs().max())
    # adjoint identity (double accumulation of the two inner products)
    W = torch.rand(r1.shape, device="cuda", generator=g)
    cg = c1.clone().requires_grad_(True)
    rg, _, _ = rasterization(*args, cg, **kw)
    (rg * W).sum().backward()
    lhs = float((rg.detach().double() * W.double()).sum())
    rhs = float((cg.grad.double() * c1.double()).sum())
    assert abs(lhs - rhs) <= 1e-4 * abs(lhs), (lhs, rhs)


def test_degenerate_inputs_do_not_crash():
    """Zero splats, nothing visible (packed and unpacked), image sizes that are not multiples of the tile size, odd tile
    sizes, a single splat: shapes are right, everything is finite, empty scenes render (background) zeros."""
    from gscodec_studio_amd import rasterization

    fx = garden(500, scale_mult=5.0)
    W, H = 100, 70
    V, K = T(fx["viewmats"][:2]), T(fx["Ks"][:2]).clone()
    K[:, :2] *= W / fx["width"]
    m, q, s, o, c = (T(fx[k]) for k in ("means", "quats", "scales", "opacities", "rgb"))

    def run(means, quats, scales, opac, colors, **kw):
        ps = [t.clone().requires_grad_(True) for t in (means, quats, scales, opac, colors)]
        rc, ra, meta = rasterization(*ps, V, K, W, H, **kw)
        (rc.sum() + ra.sum()).backward()
        return rc, ra, meta, ps

    for packed in (False, True):
        rc, ra, meta, ps = run(m + 1000.0, q, s, o, c, packed=packed)  # everything behind the cameras
        assert meta["flatten_ids"].numel() == 0 and float(ra.abs().max()) == 0.0 and float(rc.abs().max()) == 0.0
        assert all(float(p.grad.abs().max()) == 0.0 for p in ps)
    z = lambda *sh: torch.zeros(*sh, device="cuda")  # noqa: E731
    rc, ra, meta, ps = run(z(0, 3), z(0, 4), z(0, 3), z(0), z(0, 3), packed=False)
    assert rc.shape == (2, H, W, 3) and float(ra.abs().max()) == 0.0
    bg = torch.rand(2, 3, device="cuda")
    for ts in (8, 12, 16):
        rc, ra, meta, ps = run(m, q, s, o, c, packed=False, tile_size=ts, backgrounds=bg, absgrad=True, render_mode="RGB+ED")
        assert rc.shape == (2, H, W, 4) and bool(torch.isfinite(rc).all()) and meta["means2d"].absgrad.shape == (2, 500, 2)
        assert all(bool(torch.isfinite(p.grad).all()) for p in ps)
    rc, ra, meta, ps = run(m[:1], q[:1], s[:1] * 5, o[:1], torch.rand(1, 1, 3, device="cuda"), packed=False, sh_degree=0)
    assert bool(torch.isfinite(rc).all())


def test_broadcast_image_gradient_is_read_in_place():
    """``render.sum().backward()`` hands the compositing backward an EXPANDED scalar (all strides 0); the kernels read it
    through (pixel, channel) strides instead of a materialised [C,H,W,3] copy.  Same gradients as the dense tensor of
    ones, also for a per-channel weight (pixel stride 0, channel stride 1) and with a background."""
    from gscodec_studio_amd import rasterization

    d = _inputs(n=3000, cams=2, sh_degree=None, scale_mult=6.0)
    bg = torch.tensor([[0.2, 0.4, 0.6], [0.1, 0.1, 0.9]], device="cuda")

    def grads(loss_of):
        ps = [T(d[k]).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")]
        rc, ra, _ = rasterization(*ps, T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"], packed=False, backgrounds=bg)
        loss_of(rc).backward()
        return [N(p.grad) for p in ps]

    wc = torch.tensor([0.5, 1.0, 2.0], device="cuda")
    for sparse_loss, dense_loss in (
        (lambda rc: rc.sum(), lambda rc: (rc * torch.ones_like(rc)).sum()),
        (lambda rc: (rc * wc).sum(), lambda rc: (rc * wc.expand_as(rc).contiguous()).sum()),
    ):
        g0, g1 = grads(sparse_loss), grads(dense_loss)
        for a, b, name in zip(g0, g1, ("means", "quats", "scales", "opacities", "colors")):
            assert rel_l2(a, b) < 3e-4, (name, rel_l2(a, b))  # float atomics: run-to-run noise up to ~4e-5 on the quaternion gradient


def test_packed_single_camera_gradients_match_unpacked():
    """One camera: the packed projection backward writes its rows directly (no atomics needed) and the per-splat gathers go
    through gather_rows -- gradients must equal the unpacked pipeline's."""
    from gscodec_studio_amd import rasterization

    d = _inputs(n=3000, cams=1, sh_degree=3, scale_mult=5.0)

    def grads(packed):
        ps = [T(d[k]).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")]
        rc, ra, _ = rasterization(*ps, T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"], sh_degree=3, packed=packed)
        w = torch.linspace(0.5, 1.5, rc.numel(), device="cuda").reshape(rc.shape)
        ((rc * w).sum() + 0.3 * ra.sum()).backward()
        return N(rc), [N(p.grad) for p in ps]

    rc0, g0 = grads(False)
    rc1, g1 = grads(True)
    assert_close(rc1, rc0, 1e-4, 2e-6, "colors", max_bad_frac=1e-3)  # (the packed radius formula differs by design, quirk 1)
    for a, b, name in zip(g1, g0, ("means", "quats", "scales", "opacities", "colors")):
        assert rel_l2(a, b) < 2e-3, (name, rel_l2(a, b))


def test_inplace_edit_of_the_render_before_backward_raises():
    """The segmented backward needs the FINAL render (colour behind a segment = final - checkpoint); the output is therefore
    saved through save_for_backward, and modifying the returned image in place before backward() must raise autograd's
    version-counter error instead of silently producing wrong gradients.  Out-of-place post-processing is unaffected."""
    from gscodec_studio_amd import rasterization

    d = _inputs(n=1500, cams=1, sh_degree=None)
    P = {k: T(d[k], True) for k in ("means", "quats", "scales", "opacities", "colors")}
    args = (T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"])
    rc, ra, _ = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], *args, packed=False)
    rc.clamp_(0.0, 0.5)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        rc.sum().backward()
    rc, ra, _ = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], *args, packed=False)
    torch.clamp(rc, 0.0, 0.5).sum().backward()  # the out-of-place form works
    assert all(bool(torch.isfinite(p.grad).all()) for p in P.values())


@pytest.mark.parametrize("tile_size", [18, 24, 32])
def test_tile_sizes_above_16_as_sub_tiles(tile_size):
    """The reference launches tile_size^2 threads per tile and takes up to 32 (rasterize_to_pixels_fwd.cu:228); here the even
    sizes 18 .. 32 are composited as 2 x 2 sub-tiles of half the size (round 5): binning in the caller's tile size (meta as the
    reference's), image and gradients against the oracle run with the SAME tile size, masks and backgrounds included."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd import _wrapper as ops

    d = _inputs(n=2500, cams=2, sh_degree=None)
    P = {k: T(d[k], True) for k in ("means", "quats", "scales", "opacities", "colors")}
    rc, ra, meta = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"],
                                 packed=False, tile_size=tile_size)
    o_rc, o_ra, om = O.rasterization(d["means"], d["quats"], d["scales"], d["opacities"], d["colors"], d["viewmats"], d["Ks"], d["W"], d["H"],
                                     tile_size=tile_size)
    assert meta["tile_size"] == tile_size and tuple(meta["isect_offsets"].shape) == om["isect_offsets"].shape
    assert np.array_equal(N(meta["isect_offsets"]), om["isect_offsets"]) or (N(meta["radii"]) != om["radii"]).any()
    _, _, o_li, bl = O.rasterize_fwd(om["means2d"], om["conics"], om["colors"], om["opacities"], d["W"], d["H"], tile_size, om["isect_offsets"],
                                     om["flatten_ids"], return_borderline=True)
    ok = bl == 0
    assert_close(N(ra)[ok], o_ra[ok], 1e-4, 5e-5, "alphas", max_bad_frac=2e-4)
    assert_close(N(rc)[ok], o_rc[ok], 1e-4, 5e-5, "colors", max_bad_frac=2e-4)
    rs = np.random.RandomState(tile_size)
    v_rc = rs.randn(*o_rc.shape).astype(np.float32) * ok[..., None]
    (rc * T(v_rc)).sum().backward()
    v_m2, v_cn, v_col, v_op, _ = O.rasterize_bwd(om["means2d"], om["conics"], om["colors"], om["opacities"], d["W"], d["H"], tile_size,
                                                 om["isect_offsets"], om["flatten_ids"], o_ra, o_li, v_rc, np.zeros_like(o_ra))
    vm, _, vq, vs, _ = O.projection_bwd(d["means"], None, d["quats"], d["scales"], d["viewmats"], d["Ks"], d["W"], d["H"], 0.3, "pinhole",
                                        om["radii"], om["conics"], None, v_m2, np.zeros_like(om["depths"]), v_cn, None, need_viewmats=False)
    for name, ref in (("means", vm), ("quats", vq), ("scales", vs), ("opacities", v_op.sum(0)), ("colors", v_col.sum(0))):
        assert rel_l2(N(P[name].grad), ref) < 2e-3, (name, tile_size, rel_l2(N(P[name].grad), ref))
    # the op itself, with tile masks and a background, on the oracle's own binning
    th, tw = om["isect_offsets"].shape[1:]
    masks = rs.rand(2, th, tw) > 0.3
    bg = rs.rand(2, 3).astype(np.float32)
    m_rc, m_ra, _ = O.rasterize_fwd(om["means2d"], om["conics"], om["colors"], om["opacities"], d["W"], d["H"], tile_size, om["isect_offsets"],
                                    om["flatten_ids"], backgrounds=bg, masks=masks)
    g_rc, g_ra = ops.rasterize_to_pixels(T(om["means2d"]), T(om["conics"]), T(om["colors"]), T(om["opacities"]), d["W"], d["H"], tile_size,
                                         T(om["isect_offsets"]), T(om["flatten_ids"]), backgrounds=T(bg), masks=T(masks))
    pm = np.repeat(np.repeat(masks, tile_size, 1), tile_size, 2)[:, :d["H"], :d["W"]] & ok
    assert_close(N(g_rc)[pm], m_rc[pm], 1e-4, 5e-5, "masked render (kept tiles)", max_bad_frac=2e-4)
    off = ~np.repeat(np.repeat(masks, tile_size, 1), tile_size, 2)[:, :d["H"], :d["W"]]
    assert np.allclose(N(g_rc)[off], np.broadcast_to(bg[:, None, None, :], N(g_rc).shape)[off])


def test_odd_tile_sizes_above_16_are_rejected_up_front():
    from gscodec_studio_amd import rasterization

    d = _inputs(n=100, cams=1, sh_degree=None)
    for ts in (17, 33):
        with pytest.raises(AssertionError, match="tile_size"):
            rasterization(T(d["means"]), T(d["quats"]), T(d["scales"]), T(d["opacities"]), T(d["colors"]), T(d["viewmats"]), T(d["Ks"]),
                          d["W"], d["H"], packed=False, tile_size=ts)


def test_repeated_backward_and_means_gradient_routes():
    """(1) The compositing backward's gradient rows are zero-filled by the forward launch and consumed by the first
    backward: a second backward over a retained graph must fill its own and give the same gradients.
    (2) d/d means through the SH view directions is added inside the projection's backward kernel (the means are routed
    through the projection node): the total equals the sum of the two routes taken separately."""
    from gscodec_studio_amd import rasterization

    d = _inputs(n=2500, cams=2, sh_degree=3)
    P = [T(d[k]).requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "colors")]
    rc, ra, _ = rasterization(*P, T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"], sh_degree=3, packed=False)
    w = torch.rand_like(rc)
    g1 = torch.autograd.grad((rc * w).sum() + ra.sum(), P, retain_graph=True)
    g2 = torch.autograd.grad((rc * w).sum() + ra.sum(), P)
    for a, b, name in zip(g1, g2, ("means", "quats", "scales", "opacities", "sh")):
        assert rel_l2(N(a), N(b)) < 1e-4, name  # (float atomics: the order differs run to run; quaternion gradients reach ~2e-5)
    # the same on a 32x16 crop: two tiles for 5000 gradient rows, i.e. more than a tile workgroup takes on as its side job --
    # the forward zero-fills them with a plain fill instead
    rc_s, ra_s, meta_s = rasterization(*P, T(d["viewmats"]), T(d["Ks"]), 32, 16, sh_degree=3, packed=False)
    assert int((meta_s["tiles_per_gauss"] > 0).sum()) > 0
    h1 = torch.autograd.grad(rc_s.sum() + ra_s.sum(), P, retain_graph=True)
    h2 = torch.autograd.grad(rc_s.sum() + ra_s.sum(), P)
    for a, b, name in zip(h1, h2, ("means", "quats", "scales", "opacities", "sh")):
        assert rel_l2(N(a), N(b)) < 1e-4, name
    untouched = N((meta_s["tiles_per_gauss"] == 0).all(dim=0))
    assert (N(h1[3])[untouched] == 0).all() and (N(h1[4])[untouched] == 0).all()

    # route split: colours computed from detached means (no SH route), then from detached geometry (SH route only)
    m = P[0]
    m_geo, m_sh = m.detach().clone().requires_grad_(True), m.detach().clone().requires_grad_(True)
    from gscodec_studio_amd import rendering as R
    from gscodec_studio_amd._wrapper import spherical_harmonics_view

    rc_a, ra_a, meta = rasterization(m_geo, *[p.detach() for p in P[1:4]], P[4].detach(), T(d["viewmats"]), T(d["Ks"]),
                                     d["W"], d["H"], sh_degree=3, packed=False)
    # same scene, gradient w.r.t. means through the projection only: feed the colours in pre-evaluated
    cols = spherical_harmonics_view(3, m_sh, T(d["viewmats"]), P[4].detach(), meta["radii"])
    rc_b, ra_b, _ = rasterization(m_geo, *[p.detach() for p in P[1:4]], cols, T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"],
                                  sh_degree=None, packed=False)
    assert_close(N(rc_b), N(rc_a), rtol=1e-5, atol=1e-6)
    gg, gs = torch.autograd.grad((rc_b * w).sum() + ra_b.sum(), [m_geo, m_sh])
    assert rel_l2(N(g1[0]), N(gg + gs)) < 1e-4
    assert float(gs.abs().max()) > 0


@pytest.mark.parametrize("sh_degree,K", [(3, 16), (2, 16), (1, 4), (0, 2)])
def test_split_sh_coefficients_match_the_concatenated_tensor(sh_degree, K):
    """Opt-in beyond the reference's signature: ``colors=(sh0, shN)`` -- the two parameters the trainers keep (reference
    examples/simple_trainer.py:779-786 concatenates them before every render) -- gives the image of ``torch.cat([sh0, shN], 1)``
    bit for bit, and the gradients of the two tensors are the two slices of the concatenated tensor's gradient.  Partial
    bands (K > (deg + 1)^2) and an N that is no multiple of the wave size exercise the unstaged store path; routes that
    cannot take the pair (packed) fall back to the cat."""
    from gscodec_studio_amd import rasterization

    d = _inputs(n=3001, cams=2, sh_degree=3)
    sh = d["colors"][:, :K].copy()
    base = [T(d[k]) for k in ("means", "quats", "scales", "opacities")]
    vm, Ks = T(d["viewmats"]), T(d["Ks"])
    w = torch.linspace(0.5, 1.5, 2 * d["H"] * d["W"] * 3, device="cuda").reshape(2, d["H"], d["W"], 3)

    def run(colors, packed=False):
        ps = [t.clone().requires_grad_(True) for t in base]
        rc, ra, _ = rasterization(*ps, colors, vm, Ks, d["W"], d["H"], sh_degree=sh_degree, packed=packed)
        ((rc * w).sum() + ra.sum()).backward()
        return rc.detach(), ra.detach(), ps

    cat = T(sh).requires_grad_(True)
    rc0, ra0, ps0 = run(cat)
    sh0, shN = T(sh[:, :1]).requires_grad_(True), T(sh[:, 1:]).requires_grad_(True)
    rc1, ra1, ps1 = run((sh0, shN))
    assert torch.equal(rc1, rc0) and torch.equal(ra1, ra0)
    # (the compositing backward adds with float atomics: its sums differ in the last bits from run to run)
    assert rel_l2(N(sh0.grad), N(cat.grad[:, :1])) < 1e-4 and rel_l2(N(shN.grad), N(cat.grad[:, 1:])) < 1e-4
    n_act = (sh_degree + 1) ** 2
    assert float(shN.grad[:, n_act - 1:].abs().max() if n_act - 1 < K - 1 else 0.0) == 0.0  # inactive bands: exact zeros
    for a, b in zip(ps1, ps0):
        assert rel_l2(N(a.grad), N(b.grad)) < 1e-4  # (float atomics in the compositing backward: run-to-run noise)
    # a route that cannot take the pair falls back to the concatenation
    sh0p, shNp = T(sh[:, :1]).requires_grad_(True), T(sh[:, 1:]).requires_grad_(True)
    rc2, _, _ = run((sh0p, shNp), packed=True)
    rc3, _, _ = run(T(sh).requires_grad_(True), packed=True)
    assert torch.equal(rc2, rc3) and sh0p.grad is not None and shNp.grad is not None


@pytest.mark.parametrize("sh_degree", [3, None])
def test_deterministic_backward_is_bit_reproducible(sh_degree):
    """``deterministic=True`` (opt-in): the compositing backward accumulates the per-splat sums in
    fixed point, so the order in which the (tile, segment) work items reach a splat no longer matters -- two runs give
    BIT-IDENTICAL gradients for every parameter (the float-atomic default differs in the low bits from run to run, like the
    reference), and they agree with the default to fp32 rounding.  Exercised with long lists (several segments per tile,
    solo and cooperative tiles), absgrad and a background."""
    from gscodec_studio_amd import rasterization

    d = _inputs(n=4000, cams=2, sh_degree=sh_degree, scale_mult=12.0)
    names = ("means", "quats", "scales", "opacities", "colors")
    bg = torch.tensor([[0.2, 0.4, 0.6], [0.1, 0.1, 0.9]], device="cuda")
    w = torch.linspace(0.5, 1.5, 2 * d["H"] * d["W"] * 3, device="cuda").reshape(2, d["H"], d["W"], 3)

    def run(det):
        ps = [T(d[k]).requires_grad_(True) for k in names]
        rc, ra, meta = rasterization(*ps, T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"], sh_degree=sh_degree, packed=False,
                                     backgrounds=bg, absgrad=True, deterministic=det)
        meta["means2d"].retain_grad()
        ((rc * w).sum() + 0.3 * ra.sum()).backward()
        return [p.grad.clone() for p in ps] + [meta["means2d"].grad.clone(), meta["means2d"].absgrad.clone()]

    a, b = run(True), run(True)
    for x, y, name in zip(a, b, names + ("means2d", "absgrad")):
        assert torch.equal(x, y), f"{name}: deterministic runs differ"
    c = run(False)
    for x, y, name in zip(a, c, names + ("means2d", "absgrad")):
        assert rel_l2(N(x), N(y)) < 5e-5, (name, rel_l2(N(x), N(y)))


@pytest.mark.parametrize("mode", ["sh", "sh_split", "colors"])
def test_prefilled_gradients_equal_full_writes(mode):
    """``GradPrefill``: the per-gaussian gradients of the projection node are allocated behind the compositing gradient rows,
    zero-filled by the compositing FORWARD, and its backward stores the visible gaussians' rows only.  A second backward
    through the same graph (retain_graph) finds the hand-over consumed and takes the path that allocates and writes every
    row itself: with ``deterministic=True`` both must give bit-identical gradients for every parameter, and a gaussian no
    camera sees must get exact zeros from both."""
    from gscodec_studio_amd import _wrapper as W
    from gscodec_studio_amd import rasterization

    if not W.PREFILL_ENABLED:
        pytest.skip("GS_GRAD_PREFILL=0: the prefilled hand-over this test is about is switched off")
    sh_degree = None if mode == "colors" else 3
    d = _inputs(n=5000, cams=2, sh_degree=sh_degree, scale_mult=6.0)
    d["means"] = d["means"].copy()
    d["means"][::3] += 100.0  # a third of the gaussians far outside every frustum
    names = ["means", "quats", "scales", "opacities"]
    ps = [T(d[k]).requires_grad_(True) for k in names]
    if mode == "sh_split":
        sh = T(d["colors"])
        cols = [sh[:, :1].clone().requires_grad_(True), sh[:, 1:].clone().requires_grad_(True)]
        colors = (cols[0], cols[1])
    else:
        cols = [T(d["colors"]).requires_grad_(True)]
        colors = cols[0]
    w = torch.linspace(0.5, 1.5, 2 * d["H"] * d["W"] * 3, device="cuda").reshape(2, d["H"], d["W"], 3)
    rc, ra, meta = rasterization(ps[0], ps[1], ps[2], ps[3], colors, T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"],
                                 sh_degree=sh_degree, packed=False, deterministic=True)
    loss = (rc * w).sum() + 0.3 * ra.sum()
    loss.backward(retain_graph=True)
    first = [p.grad.clone() for p in ps + cols]
    # the gradients handed out by the first backward are views of ONE buffer behind the compositing gradient rows
    assert len({p.grad.untyped_storage().data_ptr() for p in ps + cols}) == 1
    for p in ps + cols:
        p.grad = None
    loss.backward()
    second = [p.grad.clone() for p in ps + cols]
    assert len({p.grad.untyped_storage().data_ptr() for p in ps + cols}) > 1
    unseen = (meta["radii"] <= 0).all(dim=0)
    assert int(unseen.sum()) >= 5000 // 3 and int((~unseen).sum()) > 500
    for a, b, name in zip(first, second, names + ["colors0", "colors1"]):
        assert torch.equal(a, b), f"{name}: prefilled and fully written gradients differ"
        assert float(a[unseen].abs().max()) == 0.0, name
        assert float(a.abs().max()) > 0.0, name


@pytest.mark.parametrize("split", [False, True])
def test_fused_sh_projection_backward_equals_two_launches(split, monkeypatch):
    """The SH backward fused behind the projection backward (one launch, `gs_projection_rows_bwd(sh_coeffs=...)`) against the
    two-launch form (`gs_sh_view_bwd`, then `gs_projection_rows_bwd(v_means_add=...)`): the same per-lane arithmetic from the
    same (deterministically accumulated) gradient rows; the two kernels are compiled separately, so fma contraction may differ
    in the last bit of the direction gradient -- everything agrees to 1e-6 (relative L2), most tensors bit for bit."""
    from gscodec_studio_amd import _wrapper as W
    from gscodec_studio_amd import rasterization

    d = _inputs(n=5000, cams=2, sh_degree=3, scale_mult=6.0)
    w = torch.linspace(0.5, 1.5, 2 * d["H"] * d["W"] * 3, device="cuda").reshape(2, d["H"], d["W"], 3)

    def run(fused):
        monkeypatch.setattr(W, "_FUSE_SH_BWD", fused)
        ps = [T(d[k]).requires_grad_(True) for k in ("means", "quats", "scales", "opacities")]
        sh = T(d["colors"])
        cols = [sh[:, :1].clone().requires_grad_(True), sh[:, 1:].clone().requires_grad_(True)] if split else [sh.clone().requires_grad_(True)]
        rc, ra, _ = rasterization(*ps, tuple(cols) if split else cols[0], T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"], sh_degree=3,
                                  packed=False, deterministic=True)
        ((rc * w).sum() + 0.3 * ra.sum()).backward()
        return [p.grad.clone() for p in ps + cols]

    a, b = run(True), run(False)
    for x, y, name in zip(a, b, ["means", "quats", "scales", "opacities", "sh0", "shN"]):
        assert rel_l2(N(x), N(y)) < 1e-6, (name, rel_l2(N(x), N(y)))
        assert float(x.abs().max()) > 0


@pytest.mark.parametrize("subset", [("means",), ("colors",), ("quats", "opacities"), ("means", "scales", "colors")])
def test_partial_requires_grad_matches_full(subset):
    """Only SOME inputs require a gradient (frozen SH, frozen geometry, ...): the prefilled-gradient hand-over, the fused
    SH + projection backward and their fall-backs must give those inputs the gradients of the all-inputs run (deterministic
    compositing backward: agreement to 1e-6), and leave the others without one."""
    from gscodec_studio_amd import rasterization

    d = _inputs(n=4000, cams=2, sh_degree=3, scale_mult=6.0)
    names = ("means", "quats", "scales", "opacities", "colors")
    w = torch.linspace(0.5, 1.5, 2 * d["H"] * d["W"] * 3, device="cuda").reshape(2, d["H"], d["W"], 3)

    def run(req):
        ps = [T(d[k]).requires_grad_(k in req) for k in names]
        rc, ra, _ = rasterization(*ps, T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"], sh_degree=3, packed=False, deterministic=True)
        ((rc * w).sum() + 0.3 * ra.sum()).backward()
        return {k: (p.grad.clone() if p.grad is not None else None) for k, p in zip(names, ps)}

    full, part = run(names), run(subset)
    for k in names:
        if k in subset:
            assert part[k] is not None and rel_l2(N(part[k]), N(full[k])) < 1e-6, (k, rel_l2(N(part[k]), N(full[k])))
        else:
            assert part[k] is None, k


@pytest.mark.parametrize("sh_degree", [None, 3])
def test_empty_and_invisible_inputs(sh_degree):
    """Edge cases of the row pipeline (reference tests/test_basic.py keeps none; its kernels return empty tensors): zero
    gaussians, and gaussians that no camera sees (everything behind the cameras) -- forward gives the background, every
    gradient is defined and exactly zero, `meta` keeps its shapes."""
    from gscodec_studio_amd import rasterization

    d = _inputs(n=500, cams=2, sh_degree=sh_degree)
    bg = torch.tensor([[0.2, 0.4, 0.6], [0.1, 0.1, 0.9]], device="cuda")
    for case in ("none_visible", "zero_gaussians"):
        sel = slice(0, 0) if case == "zero_gaussians" else slice(None)
        means = d["means"].copy()[sel]
        if case == "none_visible":
            means = means + 1000.0  # far outside every frustum
        ps = [T(means).requires_grad_(True)] + [T(d[k][sel]).requires_grad_(True) for k in ("quats", "scales", "opacities", "colors")]
        rc, ra, meta = rasterization(*ps, T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"], sh_degree=sh_degree, packed=False, backgrounds=bg)
        assert rc.shape == (2, d["H"], d["W"], 3) and ra.shape == (2, d["H"], d["W"], 1)
        assert float(ra.detach().abs().max()) == 0.0
        assert torch.equal(rc, bg[:, None, None, :].expand_as(rc)), case
        assert meta["radii"].shape == (2, means.shape[0]) and int(meta["flatten_ids"].numel()) == 0
        assert int(meta["isect_offsets"].abs().max()) == 0
        (rc.sum() + ra.sum()).backward()
        for p in ps:
            assert p.grad is not None and p.grad.shape == p.shape and (p.numel() == 0 or float(p.grad.abs().max()) == 0.0), case


@pytest.mark.parametrize("mode", ["RGB+D", "RGB+ED", "ED", "D"])
@pytest.mark.parametrize("colors_kind", ["sh", "rgb"])
def test_depth_modes_on_the_rows_route_equal_the_cat_route(mode, colors_kind):
    """RGB+D / RGB+ED with the colours in the splat rows: the four channels are a VIEW of the rows (column 9 is the depth) and the
    expected-depth tail is one kernel -- against the reference's own formulation (cat of colours and depths, slice / clamp / div /
    cat on the image: gsplat/rendering.py:418-424, 471-477) evaluated through the separate-array route (packed projection outputs)."""
    from gscodec_studio_amd import rasterization
    from util import garden

    fx = garden(5000)
    n = fx["means"].shape[0]
    rs = np.random.RandomState(4)
    cols = (rs.randn(n, 16, 3) * 0.3).astype(np.float32) if colors_kind == "sh" else fx["rgb"].astype(np.float32)
    kw = dict(sh_degree=3) if colors_kind == "sh" else {}
    vm, Ks, W, H = T(fx["viewmats"][:2]), T(fx["Ks"][:2]), fx["width"], fx["height"]
    outs = []
    for route in ("rows", "reference"):
        P = [T(fx[k], True) for k in ("means", "quats", "scales", "opacities")] + [T(cols, True)]
        if route == "rows":
            rc, ra, meta = rasterization(*P, vm, Ks, W, H, packed=False, render_mode=mode, **kw)
        else:
            # the reference's composition on top of the RGB call's pieces: render colours and depths as channels, then the ED tail in torch
            rc_, ra, meta = rasterization(*P, vm, Ks, W, H, packed=True, render_mode=mode.replace("ED", "D"), **kw)
            rc = torch.cat([rc_[..., :-1], rc_[..., -1:] / ra.clamp(min=1e-10)], dim=-1) if mode.endswith("ED") else rc_
        wgt = torch.linspace(0.5, 1.5, rc.numel(), device=rc.device).view_as(rc)
        ((rc * wgt).sum() + 0.3 * ra.sum()).backward()
        outs.append((rc.detach(), ra.detach(), [None if p.grad is None else p.grad.clone() for p in P]))
    (rc0, ra0, g0), (rc1, ra1, g1) = outs
    assert rc0.shape == rc1.shape and rc0.shape[-1] == {"RGB+D": 4, "RGB+ED": 4, "ED": 1, "D": 1}[mode]
    assert_close(N(rc0), N(rc1), 1e-4, 1e-5, f"{mode} render", max_bad_frac=2e-4)
    assert_close(N(ra0), N(ra1), 1e-4, 1e-5, f"{mode} alphas", max_bad_frac=2e-4)
    for name, a, b in zip(("means", "quats", "scales", "opacities", "colors"), g0, g1):
        if a is None or b is None:  # (depth-only modes: the colours take no part)
            assert name == "colors" and mode in ("D", "ED") and (a is None or float(a.abs().max()) == 0.0) and (b is None or float(b.abs().max()) == 0.0)
            continue
        assert rel_l2(N(a), N(b)) < 2e-3, (mode, colors_kind, name, rel_l2(N(a), N(b)))


def test_expected_depth_kernels_equal_the_torch_formulation():
    from gscodec_studio_amd.rendering import _ExpectedDepth

    rs = np.random.RandomState(0)
    for ch in (1, 4):
        r = T(rs.rand(2, 37, 41, ch).astype(np.float32), True)
        a_np = rs.rand(2, 37, 41, 1).astype(np.float32)
        a_np[0, :3, :5] = 0.0          # empty pixels: the clamp's flat side
        a_np[1, 5, 7] = 1e-10
        a = T(a_np, True)
        v = T(rs.randn(2, 37, 41, ch).astype(np.float32))
        out = _ExpectedDepth.apply(r, a)
        (out * v).sum().backward()
        r2, a2 = T(N(r), True), T(a_np, True)
        ref = torch.cat([r2[..., :-1], r2[..., -1:] / a2.clamp(min=1e-10)], dim=-1)
        (ref * v).sum().backward()
        assert torch.equal(out, ref)
        assert torch.equal(r.grad, r2.grad)
        assert_close(N(a.grad), N(a2.grad), 1e-6, 0.0, "d/d alpha")
