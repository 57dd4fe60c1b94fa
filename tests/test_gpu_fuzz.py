"""Randomised differential test of the compositing kernels against the CPU oracle: image sizes that are not multiples of
the tile, tile sizes 4..16, 1..32 channels, 1..3 cameras, backgrounds / tile masks on and off, dense and sparse scenes,
saturating opacities (early termination) and depth ties -- configurations the hand-written cases do not enumerate."""
import math
import os

import numpy as np
import pytest
import torch

from util import ROUTES, N, T, assert_close, rel_l2, tuned

pytestmark = pytest.mark.gpu


def _scene(rs):
    C = int(rs.randint(1, 4))
    W, H = int(rs.randint(17, 260)), int(rs.randint(17, 200))
    ts = int(rs.choice([4, 8, 12, 16]))
    n = int(rs.choice([40, 300, 1500, 4000]))
    D = int(rs.choice([1, 2, 3, 4, 5, 9, 8, 12, 16, 17, 32]))  # (round 5: the wide instances and the two-halves route)
    means2d = (rs.rand(C, n, 2) * np.array([W + 30, H + 30]) - 15).astype(np.float32)
    # random PSD 2x2 covariances (pixels^2), conic = inverse
    s1 = np.exp(rs.uniform(np.log(0.6), np.log(rs.choice([6.0, 25.0])), size=(C, n)))
    s2 = s1 * np.exp(rs.uniform(-1.2, 1.2, size=(C, n)))
    th = rs.uniform(0, np.pi, size=(C, n))
    c, s = np.cos(th), np.sin(th)
    a = c * c * s1 * s1 + s * s * s2 * s2
    b = c * s * (s1 * s1 - s2 * s2)
    d = s * s * s1 * s1 + c * c * s2 * s2
    det = a * d - b * b
    conics = np.stack([d / det, -b / det, a / det], -1).astype(np.float32)
    radii = np.ceil(3.0 * np.sqrt(np.maximum(a, d))).astype(np.int32)
    radii[rs.rand(C, n) < 0.1] = 0
    depths = np.round(rs.rand(C, n) * 20 + 0.2, int(rs.choice([1, 4]))).astype(np.float32)  # 1 decimal: many exact ties
    opac = rs.rand(C, n).astype(np.float32) ** (0.3 if rs.rand() < 0.5 else 2.0)  # mostly opaque / mostly faint
    colors = rs.rand(C, n, D).astype(np.float32)
    bg = rs.rand(C, D).astype(np.float32) if rs.rand() < 0.5 else None
    tw, thh = math.ceil(W / ts), math.ceil(H / ts)
    masks = (rs.rand(C, thh, tw) > 0.25) if rs.rand() < 0.3 else None
    return dict(C=C, W=W, H=H, ts=ts, n=n, D=D, means2d=means2d, conics=conics, radii=radii, depths=depths, opac=opac,
                colors=colors, bg=bg, masks=masks, tw=tw, th=thh)


# GS_FUZZ_SEED_OFFSET=k shifts every scene's seed: extra fuzz sessions beyond the 24 + 16 scenes of the default suite
_SEED_OFFSET = int(os.environ.get("GS_FUZZ_SEED_OFFSET", "0"))

# every kernel route the tuning knobs can select runs in the suite: the default over 24 scenes, the others over 8 each
_FUZZ = [("default", s) for s in range(24)] + [(r, s) for r in ROUTES if r != "default" for s in range(8)]


@pytest.mark.parametrize("route,seed", _FUZZ)
def test_compositing_fuzz_vs_oracle(route, seed):
    with tuned(route):
        _compositing_fuzz(seed)


def _compositing_fuzz(seed):
    from oracle import gs_oracle as O

    from gscodec_studio_amd import _wrapper as ops

    rs = np.random.RandomState(1000 + seed + _SEED_OFFSET)
    c = _scene(rs)
    tpg, ids, flat = O.isect_tiles(c["means2d"], c["radii"], c["depths"], c["ts"], c["tw"], c["th"])
    offs = O.isect_offset_encode(ids, c["C"], c["tw"], c["th"])
    # binning through the library must agree bit for bit (tile sizes / image sizes off the beaten path)
    tpg_g, ids_g, flat_g = ops.isect_tiles(T(c["means2d"]), T(c["radii"]), T(c["depths"]), c["ts"], c["tw"], c["th"])
    assert np.array_equal(N(tpg_g), tpg) and np.array_equal(N(ids_g), ids) and np.array_equal(N(flat_g), flat), "binning"
    assert np.array_equal(N(ops.isect_offset_encode(ids_g, c["C"], c["tw"], c["th"])), offs)

    o_rc, o_ra, o_li, bl = O.rasterize_fwd(c["means2d"], c["conics"], c["colors"], c["opac"], c["W"], c["H"], c["ts"], offs, flat,
                                           backgrounds=c["bg"], masks=c["masks"], return_borderline=True)
    m2, cn, col, op = T(c["means2d"], True), T(c["conics"], True), T(c["colors"], True), T(c["opac"], True)
    bg_t = T(c["bg"], True) if c["bg"] is not None else None
    rc, ra = ops.rasterize_to_pixels(m2, cn, col, op, c["W"], c["H"], c["ts"], T(offs), T(flat), backgrounds=bg_t,
                                     masks=T(c["masks"]) if c["masks"] is not None else None)
    ok = bl == 0
    kept = 1.0  # fraction of the pixels the tile masks leave to compare
    if c["masks"] is not None:  # skipped tiles: colours = background (or 0), alphas are left unwritten by the reference
        pm = np.repeat(np.repeat(c["masks"], c["ts"], 1), c["ts"], 2)[:, :c["H"], :c["W"]]
        ok = ok & pm
        kept = float(pm.mean())
    tag = f"seed {seed}: C={c['C']} {c['W']}x{c['H']} tile {c['ts']} n={c['n']} D={c['D']} bg={c['bg'] is not None} masks={c['masks'] is not None}"
    # (most of the comparable pixels must not be borderline; a random mask over a 2 x 7 tile image may itself keep less than half
    # of them -- seed offset 4600, case 12: 48.9 % -- which says nothing about the kernels)
    if kept == 0.0:
        return  # every tile masked out: nothing to compare
    assert ok.mean() > 0.5 * kept, tag
    assert_close(N(rc)[ok], o_rc[ok], 1e-4, 2e-5, "render_colors " + tag, max_bad_frac=1e-4)
    assert_close(N(ra)[ok], o_ra[ok], 1e-4, 2e-5, "render_alphas " + tag, max_bad_frac=1e-4)

    v_rc = rs.randn(*o_rc.shape).astype(np.float32) * ok[..., None]
    v_ra = rs.randn(*o_ra.shape).astype(np.float32) * ok[..., None]
    loss = (rc * T(v_rc)).sum() + (ra * T(v_ra)).sum()
    grads = torch.autograd.grad(loss, (m2, cn, col, op))
    o = O.rasterize_bwd(c["means2d"], c["conics"], c["colors"], c["opac"], c["W"], c["H"], c["ts"], offs, flat, o_ra, o_li,
                        v_rc, v_ra, backgrounds=c["bg"], masks=c["masks"])
    d64 = None  # the float64 build of the oracle, evaluated only when the fp32 oracle and the kernels are further apart than 5e-4
    for i, (name, got, ref) in enumerate(zip(("v_means2d", "v_conics", "v_colors", "v_opacities"), grads, o[:4])):
        scale = np.abs(ref).max()
        if scale == 0:
            assert np.abs(N(got)).max() == 0, (name, tag)
            continue
        e32 = rel_l2(N(got), ref)
        if e32 < 5e-4:
            continue
        # 5e-4 is the fp32 ORACLE's error bar, not the kernels': its back-to-front T / (1 - alpha) recurrence over the whole list
        # reaches 7e-4 on some scenes (seed offset 7700, scene 15: v_conics oracle 6.9e-4, kernels 5.8e-5 against float64).  The
        # float64 oracle decides: the kernels must be within the north star's 1e-4 of it, and closer to it than the fp32 oracle is
        if d64 is None:
            with O.precision(64):
                _, q_ra, q_li, bl64 = O.rasterize_fwd(c["means2d"], c["conics"], c["colors"], c["opac"], c["W"], c["H"], c["ts"], offs, flat,
                                                      backgrounds=c["bg"], masks=c["masks"], return_borderline=True)
                assert not ((bl64 != 0) & ok).any(), ("a pixel only the float64 run flags carries upstream gradient", tag)
                d64 = O.rasterize_bwd(c["means2d"], c["conics"], c["colors"], c["opac"], c["W"], c["H"], c["ts"], offs, flat, q_ra, q_li,
                                      v_rc, v_ra, backgrounds=c["bg"], masks=c["masks"])
        e_hip, e_orc = rel_l2(N(got).astype(np.float64), d64[i]), rel_l2(ref, d64[i])
        assert e_hip < 1e-4 and e_hip < e_orc, (name, tag, "HIP vs f64", e_hip, "fp32 oracle vs f64", e_orc, "HIP vs fp32 oracle", e32)


@pytest.mark.parametrize("seed", range(16))
def test_pipeline_fuzz_vs_oracle(seed):
    """The whole unpacked forward (projection -> SH -> binning -> compositing) on random scenes, cameras and options:
    camera model, antialiasing, SH degree or plain colours, tile size, clipping planes, radius clip, image sizes."""
    from oracle import gs_oracle as O

    from gscodec_studio_amd import rasterization

    rs = np.random.RandomState(7000 + seed + _SEED_OFFSET)
    C, n = int(rs.randint(1, 4)), int(rs.choice([200, 1500, 5000]))
    W, H = int(rs.randint(33, 320)), int(rs.randint(33, 240))
    cm = str(rs.choice(["pinhole", "pinhole", "ortho", "fisheye"]))
    aa = bool(rs.rand() < 0.4)
    deg = rs.choice([None, 0, 1, 2, 3])
    deg = None if deg is None else int(deg)
    ts = int(rs.choice([8, 16]))
    means = (rs.randn(n, 3) * np.array([1.5, 1.5, 1.0])).astype(np.float32)
    quats = rs.randn(n, 4).astype(np.float32)
    scales = np.exp(rs.uniform(np.log(0.01), np.log(0.25), size=(n, 3))).astype(np.float32)
    opac = rs.rand(n).astype(np.float32)
    K = 16
    colors = (rs.rand(n, 3).astype(np.float32) if deg is None else (rs.randn(n, K, 3) * 0.3).astype(np.float32))
    viewmats = np.tile(np.eye(4, dtype=np.float32), (C, 1, 1))
    for c in range(C):  # cameras on a ring looking at the origin from z = -4
        a = rs.uniform(-0.5, 0.5)
        R = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
        viewmats[c, :3, :3] = R
        viewmats[c, :3, 3] = np.array([rs.uniform(-0.3, 0.3), rs.uniform(-0.3, 0.3), rs.uniform(3.0, 5.0)], np.float32)
    f = (0.9 * W) if cm != "ortho" else 40.0
    Ks = np.tile(np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float32), (C, 1, 1))
    near, far = (0.01, 1e10) if rs.rand() < 0.6 else (3.2, 6.0)
    rclip = 0.0 if rs.rand() < 0.7 else 2.0
    bg = rs.rand(C, 3).astype(np.float32) if rs.rand() < 0.5 else None
    tag = f"seed {seed}: C={C} n={n} {W}x{H} {cm} aa={aa} deg={deg} tile {ts} near/far {near}/{far} rclip {rclip}"

    o_rc, o_ra, om = O.rasterization(means, quats, scales, opac, colors, viewmats, Ks, W, H, near_plane=near, far_plane=far,
                                     radius_clip=rclip, sh_degree=deg, tile_size=ts, backgrounds=bg, camera_model=cm, antialiased=aa)
    rc, ra, meta = rasterization(T(means), T(quats), T(scales), T(opac), T(colors), T(viewmats), T(Ks), W, H, near_plane=near,
                                 far_plane=far, radius_clip=rclip, sh_degree=deg, tile_size=ts, packed=False,
                                 backgrounds=T(bg) if bg is not None else None, camera_model=cm,
                                 rasterize_mode="antialiased" if aa else "classic")
    # the projection's cull decisions sit on float thresholds: compare where both agree on visibility, require that they
    # almost always do
    vis_o, vis_g = om["radii"] > 0, N(meta["radii"]) > 0
    assert (vis_o != vis_g).mean() < 2e-3, tag
    if (vis_o != vis_g).any():
        return  # a flipped splat changes whole pixels; the per-stage tests cover the numerics
    assert np.abs(N(meta["radii"]).astype(np.int64) - om["radii"]).max() <= 1, tag
    same_radii = np.array_equal(N(meta["radii"]), om["radii"])
    if same_radii:
        assert np.array_equal(N(meta["tiles_per_gauss"]), om["tiles_per_gauss"]) if "tiles_per_gauss" in om else True
    d = np.abs(N(rc) - o_rc)
    bad = (d > 2e-3 + 2e-3 * np.abs(o_rc)).mean()
    assert bad < 2e-3, (tag, float(bad), float(d.max()))
    assert (np.abs(N(ra) - o_ra) > 2e-3).mean() < 2e-3, tag
