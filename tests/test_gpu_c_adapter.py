"""The `_C` adapter (gscodec_studio_amd/_c_adapter.py): every 3DGS name of the reference's pybind module
(gsplat/cuda/csrc/ext.cpp:3-92) called with the reference's positional torch-tensor signatures
(gsplat/cuda/include/bindings.h), the way the reference's own `_wrapper.py` calls them, and checked against the oracle
(forward stages) / the package's autograd operators (backward stages, themselves checked against the oracle elsewhere)."""
import math

import numpy as np
import pytest
import torch

from util import N, T, assert_close, garden, rel_l2

pytestmark = pytest.mark.gpu

from oracle import gs_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def C_():
    from gscodec_studio_amd._c_adapter import _C

    return _C


def _scene(n=3000, cams=2, scale_mult=5.0):
    fx = garden(n, scale_mult=scale_mult)
    d = {k: T(fx[k]) for k in ("means", "quats", "scales", "opacities", "rgb")}
    d["viewmats"], d["Ks"] = T(fx["viewmats"][:cams]), T(fx["Ks"][:cams])
    d["W"], d["H"], d["fx"], d["cams"], d["n"] = fx["width"], fx["height"], fx, cams, n
    return d


def test_every_reference_name_is_there(C_):
    names = ["compute_sh_fwd", "compute_sh_bwd", "quat_scale_to_covar_preci_fwd", "quat_scale_to_covar_preci_bwd", "proj_fwd",
             "proj_bwd", "world_to_cam_fwd", "world_to_cam_bwd", "fully_fused_projection_fwd", "fully_fused_projection_bwd",
             "isect_tiles", "isect_offset_encode", "rasterize_to_pixels_fwd", "rasterize_to_pixels_bwd",
             "rasterize_to_indices_in_range", "fully_fused_projection_packed_fwd", "fully_fused_projection_packed_bwd"]
    for nme in names:
        assert callable(getattr(C_, nme)), nme
    assert int(C_.CameraModelType.PINHOLE) == 0 and int(C_.CameraModelType.FISHEYE) == 2 and C_.CameraModelType.ORTHO.name == "ORTHO"
    with pytest.raises(AttributeError):
        C_.rasterize_to_pixels_fwd_2dgs  # out of scope


def test_forward_chain_through_the_adapter_vs_oracle(C_):
    """fully_fused_projection_fwd -> isect_tiles -> isect_offset_encode -> rasterize_to_pixels_fwd, called like the reference's
    _wrapper.py does (positional arguments, camera model enum), against the oracle's whole forward."""
    d = _scene()
    W, H, C = d["W"], d["H"], d["cams"]
    radii, means2d, depths, conics, comps = C_.fully_fused_projection_fwd(
        d["means"], None, d["quats"], d["scales"], d["viewmats"], d["Ks"], W, H, 0.3, 0.01, 1e10, 0.0, False, C_.CameraModelType.PINHOLE)
    assert comps is None and radii.dtype == torch.int32
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = C_.isect_tiles(means2d, radii, depths, None, None, C, 16, tw, th, True, True)
    offs = C_.isect_offset_encode(ids, C, tw, th)
    colors = d["rgb"].expand(C, -1, -1).contiguous()
    opac = d["opacities"].repeat(C, 1)
    rc, ra, last = C_.rasterize_to_pixels_fwd(means2d, conics, colors, opac, None, None, W, H, 16, offs, flat)
    fx = d["fx"]
    o_rc, o_ra, om = O.rasterization(fx["means"], fx["quats"], fx["scales"], fx["opacities"], fx["rgb"], fx["viewmats"][:C],
                                     fx["Ks"][:C], W, H)
    assert (N(radii) == om["radii"]).mean() > 0.999
    # the integer stages, bit for bit, against the oracle's binning of the SAME projected splats (the oracle's own projection rounds
    # without fused multiply-adds: a mean2d or depth that differs in the last bit may move a tile bound or swap two list entries)
    o_tpg, o_ids, o_flat = O.isect_tiles(N(means2d), N(radii), N(depths), 16, tw, th)
    assert np.array_equal(N(tpg), o_tpg) and np.array_equal(N(ids), o_ids) and np.array_equal(N(flat), o_flat)
    assert np.array_equal(N(offs), O.isect_offset_encode(o_ids, C, tw, th))
    assert_close(N(rc), o_rc, 1e-4, 5e-5, "adapter render", max_bad_frac=5e-4)
    assert_close(N(ra), o_ra, 1e-4, 5e-5, "adapter alpha", max_bad_frac=5e-4)
    # the reference's oracle helper, through the adapter: camera * H * W + pixel, list order inside a pixel
    g, p = C_.rasterize_to_indices_in_range(0, 10, torch.ones((C, H, W), device="cuda"), means2d, conics, opac, W, H, 16, offs, flat)
    assert g.dtype == torch.int64 and p.dtype == torch.int64 and g.shape == p.shape and int(p.max()) < C * H * W
    assert bool((p[1:] >= p[:-1]).all())


def test_backward_functions_through_the_adapter_vs_package_autograd(C_):
    """rasterize_to_pixels_bwd / fully_fused_projection_bwd / compute_sh_* with the reference's argument lists give what the
    package's autograd operators give (which the other test files check against the oracle)."""
    from gscodec_studio_amd import _wrapper as ops

    d = _scene(n=2500)
    W, H, C = d["W"], d["H"], d["cams"]
    m, q, s = (d[k].clone().requires_grad_(True) for k in ("means", "quats", "scales"))
    radii, means2d, depths, conics, _ = ops.fully_fused_projection(m, None, q, s, d["viewmats"], d["Ks"], W, H)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = ops.isect_tiles(means2d, radii, depths, 16, tw, th, n_cameras=C)
    offs = ops.isect_offset_encode(ids, C, tw, th)
    colors = d["rgb"].expand(C, -1, -1).contiguous().requires_grad_(True)
    opac = d["opacities"].repeat(C, 1).requires_grad_(True)
    rc, ra = ops.rasterize_to_pixels(means2d, conics, colors, opac, W, H, 16, offs, flat, absgrad=True)
    g = torch.Generator(device="cuda").manual_seed(3)
    v_rc, v_ra = torch.randn(rc.shape, device="cuda", generator=g), torch.randn(ra.shape, device="cuda", generator=g)
    means2d.retain_grad(), conics.retain_grad(), depths.retain_grad()
    ((rc * v_rc).sum() + (ra * v_ra).sum() + (depths * 0.1).sum()).backward()

    # -- compositing backward through the adapter (needs the forward's alphas / last ids: run the adapter's forward)
    rc2, ra2, last = C_.rasterize_to_pixels_fwd(means2d.detach(), conics.detach(), colors.detach(), opac.detach(), None, None, W, H, 16,
                                                offs, flat)
    assert torch.equal(rc2, rc.detach()) and torch.equal(ra2, ra.detach())
    v_abs, v_m2, v_cn, v_col, v_op = C_.rasterize_to_pixels_bwd(means2d.detach(), conics.detach(), colors.detach(), opac.detach(), None,
                                                               None, W, H, 16, offs, flat, ra2, last, v_rc, v_ra, True)
    for got, ref, name in ((v_m2, means2d.grad, "v_means2d"), (v_cn, conics.grad, "v_conics"), (v_col, colors.grad, "v_colors"),
                           (v_op, opac.grad, "v_opacities"), (v_abs, means2d.absgrad, "absgrad")):
        assert rel_l2(N(got), N(ref)) < 3e-4, (name, rel_l2(N(got), N(ref)))

    # -- projection backward through the adapter
    v_means, v_cov, v_quats, v_scales, v_view = C_.fully_fused_projection_bwd(
        d["means"], None, d["quats"], d["scales"], d["viewmats"], d["Ks"], W, H, 0.3, C_.CameraModelType.PINHOLE, radii, conics.detach(),
        None, means2d.grad.contiguous(), depths.grad.contiguous(), conics.grad.contiguous(), None, True)
    assert v_cov is None and v_view.shape == (C, 4, 4)
    for got, ref, name in ((v_means, m.grad, "v_means"), (v_quats, q.grad, "v_quats"), (v_scales, s.grad, "v_scales")):
        assert rel_l2(N(got), N(ref)) < 1e-5, (name, rel_l2(N(got), N(ref)))

    # -- SH through the adapter vs the oracle
    rs = np.random.RandomState(2)
    dirs = rs.randn(1000, 3).astype(np.float32)
    coeffs = rs.randn(1000, 16, 3).astype(np.float32)
    masks = rs.rand(1000) > 0.2
    col = C_.compute_sh_fwd(3, T(dirs), T(coeffs), T(masks))
    o_col = O.sh_fwd(3, dirs, coeffs, masks)
    assert_close(N(col)[masks], o_col[masks], 1e-5, 1e-6, "adapter sh fwd")
    v_col_np = rs.randn(1000, 3).astype(np.float32)
    v_coeffs, v_dirs = C_.compute_sh_bwd(16, 3, T(dirs), T(coeffs), T(masks), T(v_col_np), True)
    o_vc, o_vd = O.sh_bwd(3, dirs, coeffs, v_col_np, masks)
    assert rel_l2(N(v_coeffs), o_vc) < 1e-5 and rel_l2(N(v_dirs), o_vd) < 1e-4
    assert C_.compute_sh_bwd(16, 3, T(dirs), T(coeffs), None, T(v_col_np), False)[1] is None


def test_unfused_and_packed_names_through_the_adapter(C_):
    from gscodec_studio_amd import _wrapper as ops

    d = _scene(n=1500)
    W, H, C = d["W"], d["H"], d["cams"]
    cov, pre = C_.quat_scale_to_covar_preci_fwd(d["quats"], d["scales"], True, True, False)
    cov_ref, pre_ref = ops.quat_scale_to_covar_preci(d["quats"], d["scales"])
    assert torch.equal(cov, cov_ref) and torch.equal(pre, pre_ref)
    assert C_.quat_scale_to_covar_preci_fwd(d["quats"], d["scales"], True, False, True)[1] is None
    vq, vs = C_.quat_scale_to_covar_preci_bwd(d["quats"], d["scales"], torch.ones_like(cov), None, False)
    assert vq.shape == d["quats"].shape and vs.shape == d["scales"].shape and bool(torch.isfinite(vq).all())
    mc, cc = C_.world_to_cam_fwd(d["means"], cov, d["viewmats"])
    mc_ref, cc_ref = ops.world_to_cam(d["means"], cov, d["viewmats"])
    assert torch.equal(mc, mc_ref) and torch.equal(cc, cc_ref)
    vm, vc, vv = C_.world_to_cam_bwd(d["means"], cov, d["viewmats"], torch.ones_like(mc), None, True, False, True)
    assert vc is None and vm.shape == d["means"].shape and vv.shape == d["viewmats"].shape
    m2, c2 = C_.proj_fwd(mc, cc, d["Ks"], W, H, C_.CameraModelType.PINHOLE)
    m2_ref, c2_ref = ops.proj(mc, cc, d["Ks"], W, H)
    assert torch.equal(m2, m2_ref) and torch.equal(c2, c2_ref)
    vmc, vcc = C_.proj_bwd(mc, cc, d["Ks"], W, H, C_.CameraModelType.PINHOLE, torch.ones_like(m2), torch.ones_like(c2))
    assert vmc.shape == mc.shape and vcc.shape == cc.shape
    # packed
    out = C_.fully_fused_projection_packed_fwd(d["means"], None, d["quats"], d["scales"], d["viewmats"], d["Ks"], W, H, 0.3, 0.01,
                                               1e10, 0.0, True, C_.CameraModelType.PINHOLE)
    indptr, cam, gau, radii, means2d, depths, conics, comps = out
    ref = ops.fully_fused_projection(d["means"], None, d["quats"], d["scales"], d["viewmats"], d["Ks"], W, H, packed=True,
                                     calc_compensations=True)
    assert torch.equal(cam, ref[0]) and torch.equal(gau, ref[1]) and torch.equal(means2d, ref[3]) and torch.equal(comps, ref[6])
    assert int(indptr[-1]) == cam.numel() and indptr.shape == (C + 1,)
    g = C_.fully_fused_projection_packed_bwd(d["means"], None, d["quats"], d["scales"], d["viewmats"], d["Ks"], W, H, 0.3,
                                             C_.CameraModelType.PINHOLE, cam, gau, conics, comps, torch.ones_like(means2d),
                                             torch.ones_like(depths), torch.ones_like(conics), torch.ones_like(comps), False, True)
    assert g[0].shape == (cam.numel(), 3) and g[1] is None and g[4] is None and bool(torch.isfinite(g[2]).all())
