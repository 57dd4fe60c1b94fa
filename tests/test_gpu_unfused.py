"""GPU parity of the unfused public ops (gs_world_to_cam_*, gs_proj_*, gs_rasterize_indices_*) through the
C ABI: against the reference's golden vectors and the float64 oracle (1e-4 relative to the tensor's scale),
and the indices op against the oracle loops (integers: exact up to alpha-threshold ties) and against the
compositing forward it is the companion of."""
import numpy as np
import pytest
import torch

from util import N, T, assert_close, garden, golden

pytestmark = pytest.mark.gpu

from oracle import unfused_oracle as UO  # noqa: E402


def rel_ok(got, ref, tol=1e-4):
    got, ref = np.asarray(got, np.float64), np.asarray(ref, np.float64)
    bad = np.abs(got - ref) > tol * np.abs(ref) + tol * np.abs(ref).mean()
    return bad.mean()


def test_world_to_cam_fwd_bwd():
    from gscodec_studio_amd import world_to_cam

    gd = golden("unfused.npz")
    means, covars, viewmats = (T(gd[k]).requires_grad_(True) for k in ("means", "covars", "viewmats"))
    mc, cc = world_to_cam(means, covars, viewmats)
    assert rel_ok(N(mc), gd["w2c.means_c"]) == 0 and rel_ok(N(cc), gd["w2c.covars_c"]) == 0
    ((mc * T(gd["w2c.v_means_c"])).sum() + (cc * T(gd["w2c.v_covars_c"])).sum()).backward()
    assert rel_ok(N(means.grad), gd["w2c.v_means"]) < 1e-3
    assert rel_ok(N(covars.grad), gd["w2c.v_covars"]) < 1e-3
    ref = gd["w2c.v_viewmats"]
    assert np.abs(N(viewmats.grad) - ref).max() <= 1e-4 * np.abs(ref).max()
    # selective gradients: only the poses
    means2, covars2 = T(gd["means"]), T(gd["covars"])
    vm = T(gd["viewmats"]).requires_grad_(True)
    mc, cc = world_to_cam(means2, covars2, vm)
    (mc.sum() + cc.sum()).backward()
    assert vm.grad is not None and float(vm.grad[:, 3].abs().max()) == 0.0


@pytest.mark.parametrize("model", ["pinhole", "ortho", "fisheye"])
def test_proj_fwd_bwd(model):
    from gscodec_studio_amd import proj

    gd = golden("unfused.npz")
    means = T(gd["proj.means"]).requires_grad_(True)
    covars = T(gd["proj.covars"]).requires_grad_(True)
    W, H = int(gd["width"]), int(gd["height"])
    m2, c2 = proj(means, covars, T(gd["Ks"]), W, H, model)
    assert rel_ok(N(m2), gd[f"proj.{model}.means2d"]) < 1e-3
    assert rel_ok(N(c2), gd[f"proj.{model}.covars2d"]) < 1e-3
    ((m2 * T(gd["proj.v_means2d"])).sum() + (c2 * T(gd["proj.v_covars2d"])).sum()).backward()
    assert rel_ok(N(means.grad), gd[f"proj.{model}.v_means"], 2e-4) < 2e-3
    assert rel_ok(N(covars.grad), gd[f"proj.{model}.v_covars"], 2e-4) < 2e-3
    # float64 oracle as arbiter
    Ks = torch.tensor(gd["Ks"], dtype=torch.float64)
    (om, oc), og = UO.with_grads(lambda a, b: UO.proj(a, b, Ks, W, H, model), (gd["proj.means"], gd["proj.covars"]),
                                 (gd["proj.v_means2d"], gd["proj.v_covars2d"]))
    assert rel_ok(N(c2), oc) < 1e-3 and rel_ok(N(means.grad), og[0], 2e-4) < 2e-3


def _scene(n=400, cams=2):
    from gscodec_studio_amd import rasterization

    fx = garden(n, scale_mult=5.0)
    d = dict(means=T(fx["means"]), quats=T(fx["quats"]), scales=T(fx["scales"]), opacities=T(fx["opacities"]), colors=T(fx["rgb"]),
             viewmats=T(fx["viewmats"][:cams]), Ks=T(fx["Ks"][:cams]))
    W, H = fx["width"] // 4, fx["height"] // 4
    Ks = d["Ks"].clone()
    Ks[:, :2] /= 4
    rc, ra, meta = rasterization(d["means"], d["quats"], d["scales"], d["opacities"], d["colors"], d["viewmats"], Ks, W, H, packed=False)
    return rc, ra, meta, W, H, cams, n, d


def test_indices_in_range_vs_oracle_and_forward():
    from gscodec_studio_amd import rasterize_to_indices_in_range

    rc, ra, meta, W, H, C, n, d = _scene()
    ts = meta["tile_size"]
    args = (meta["means2d"], meta["conics"], meta["opacities"], W, H, ts, meta["isect_offsets"], meta["flatten_ids"])
    T0 = torch.ones((C, H, W), device="cuda")
    g, p, c = rasterize_to_indices_in_range(0, 10**10, T0, *args)
    assert g.dtype == torch.int64 and p.dtype == torch.int64 and c.dtype == torch.int64
    og, op, oc = UO.rasterize_to_indices_in_range(0, 10**10, N(T0), N(meta["means2d"]), N(meta["conics"]), N(meta["opacities"]), W, H, ts,
                                                   N(meta["isect_offsets"]), N(meta["flatten_ids"]))
    # per-pixel counts agree except where an alpha / transmittance threshold is decided by the last ulp of exp
    key_g = N(c) * (H * W) + N(p)
    key_o = oc * (H * W) + op
    cnt_g = np.bincount(key_g, minlength=C * H * W)
    cnt_o = np.bincount(key_o, minlength=C * H * W)
    same = cnt_g == cnt_o
    assert same.mean() > 0.999, float(same.mean())
    # pixel-major order, list order inside a pixel: identical sequences on the agreeing pixels
    assert np.all(np.diff(key_g) >= 0)
    ok_g, ok_o = same[key_g], same[key_o]
    assert np.array_equal(N(g)[ok_g], og[ok_o])
    # companion of the compositing forward: alpha = 1 - prod(1 - alpha_i) over the listed pairs
    m2, cn, opa = N(meta["means2d"]).reshape(-1, 2), N(meta["conics"]).reshape(-1, 3), N(meta["opacities"]).reshape(-1)
    flat_g = N(c) * n + N(g)
    px = (N(p) % W) + 0.5
    py = (N(p) // W) + 0.5
    dx, dy = m2[flat_g, 0] - px, m2[flat_g, 1] - py
    sig = 0.5 * (cn[flat_g, 0] * dx * dx + cn[flat_g, 2] * dy * dy) + cn[flat_g, 1] * dx * dy
    al = np.minimum(0.999, opa[flat_g] * np.exp(-sig))
    logT = np.zeros(C * H * W)
    np.add.at(logT, key_g, np.log1p(-al.astype(np.float64)))
    alpha = 1.0 - np.exp(logT)
    assert_close(alpha.reshape(C, H, W, 1), N(ra), 1e-4, 1e-5, "alpha from listed pairs vs render_alphas", max_bad_frac=1e-3)
    # batch ranges: [0,1) then [1, inf) with the transmittance reached after the first batch == the full list
    g0, p0, c0 = rasterize_to_indices_in_range(0, 1, T0, *args)
    k0 = N(c0) * (H * W) + N(p0)
    f0 = N(c0) * n + N(g0)
    dx0, dy0 = m2[f0, 0] - ((N(p0) % W) + 0.5), m2[f0, 1] - ((N(p0) // W) + 0.5)
    s0 = 0.5 * (cn[f0, 0] * dx0 * dx0 + cn[f0, 2] * dy0 * dy0) + cn[f0, 1] * dx0 * dy0
    a0 = np.minimum(0.999, opa[f0] * np.exp(-s0)).astype(np.float32)
    T1 = np.ones(C * H * W, np.float32)
    for k, a in zip(k0, a0):  # sequential product in list order, fp32 like the kernel
        T1[k] = T1[k] * (np.float32(1.0) - a)
    g1, p1, c1 = rasterize_to_indices_in_range(1, 10**10, T(T1.reshape(C, H, W)), *args)
    assert abs((len(g0) + len(g1)) - len(g)) <= max(2, int(1e-3 * len(g)))


def test_indices_in_range_empty_and_asserts():
    from gscodec_studio_amd import rasterize_to_indices_in_range

    C, n, W, H, ts = 1, 10, 32, 32, 16
    z = torch.zeros
    g, p, c = rasterize_to_indices_in_range(0, 5, torch.ones(C, H, W, device="cuda"), z(C, n, 2, device="cuda"), z(C, n, 3, device="cuda"),
                                            z(C, n, device="cuda"), W, H, ts, z(C, 2, 2, dtype=torch.int32, device="cuda"),
                                            z(0, dtype=torch.int32, device="cuda"))
    assert len(g) == 0 and len(p) == 0 and len(c) == 0
    with pytest.raises(AssertionError):
        rasterize_to_indices_in_range(0, 5, torch.ones(C, H, W, device="cuda"), z(C, n, 2, device="cuda"), z(C, n, 3, device="cuda"),
                                      z(C, n, device="cuda"), W, H, 8, z(C, 2, 2, dtype=torch.int32, device="cuda"),
                                      z(4, dtype=torch.int32, device="cuda"))
