"""The temporal slice (and, opt-in, the activations + round quantizer hooks) evaluated INSIDE the projection kernels
(``rasterization(dynamic=...)``, csrc/projection_dyn.hip; SURVEY 8f rank 2 "folded into the projection kernel's load phase") against
the same chain through the stand-alone operators: ``STE`` hooks -> activations -> ``temporal_slice`` -> ``rasterization``
(reference examples/simple_trainer_dyngs.py:463-554).

* the fused FORWARD is bit-identical to the chain: radii, splat rows, depths, the binning's integer stages and the image;
* the fused BACKWARD agrees to 1e-6 (relative L2: the shared VJP code is left contractible for accuracy, csrc/gs_common.h GS_FP_STRICT,
  and the compiler fuses it differently in the two kernels)
  given the same gradient rows (operator level) and through ``rasterization`` with the deterministic compositing backward; on the default
  route (float atomics in the compositing backward) within its run-to-run noise;
* the quantizer clamps the parameters in place exactly as the hooks do;
* the routes the fused kernels do not cover (packed, SH colours, camera-pose gradients) take the chain themselves.
The chain itself is pinned elsewhere: tests/test_gpu_dynamic.py (slice vs the reference trainer's own statements),
tests/test_gpu_quantize.py (STE vs the reference's ops.py), tests/test_gpu_configs.py (config 5 vs the oracle chain)."""
import numpy as np
import pytest
import torch

from util import N, T, assert_close, garden, rel_l2

pytestmark = pytest.mark.gpu

KEYS = ("means", "scales", "quats", "opacities", "trbf_center", "trbf_scale", "motion", "omega", "colors")
BDS = dict(scales=(-10.0, 2.0, 8), quats=(-1.0, 1.0, 8), opacities=(-7.0, 7.0, 8), colors=(-7.5, 7.5, 8))


def _params(n, seed=0, channels=3, activated=True):
    """Raw trainer parameters (log-scales, logits, log trbf_scale) or, with ``activated``, the values the slice takes."""
    fx = garden(n, scale_mult=5.0)
    n = fx["means"].shape[0]
    rs = np.random.RandomState(seed)
    op = np.clip(fx["opacities"], 1e-4, 1 - 1e-4)
    raw = dict(
        means=fx["means"].astype(np.float32), scales=np.log(np.maximum(fx["scales"], 1e-6)).astype(np.float32),
        quats=fx["quats"].astype(np.float32), opacities=np.log(op / (1 - op)).astype(np.float32),
        trbf_center=rs.uniform(0, 1, (n, 1)).astype(np.float32), trbf_scale=rs.uniform(-1.5, 0.5, (n, 1)).astype(np.float32),
        motion=(0.02 * rs.randn(n, 9)).astype(np.float32), omega=(0.1 * rs.randn(n, 4)).astype(np.float32),
        colors=(fx["rgb"] if channels == 3 else rs.rand(n, channels)).astype(np.float32))
    if activated:
        raw["scales"], raw["trbf_scale"] = np.exp(raw["scales"]), np.exp(raw["trbf_scale"])
        raw["opacities"] = (1 / (1 + np.exp(-raw["opacities"].astype(np.float64)))).astype(np.float32)
    return fx, raw


def _P(raw):
    return {k: torch.nn.Parameter(T(raw[k])) for k in KEYS}


def _render(P, fx, t, C=1, fused=True, raw=(), quantize=None, **kw):
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd.dynamic import DynamicSlice

    vm, Ks = T(fx["viewmats"][:C]), T(fx["Ks"][:C])
    ds = DynamicSlice(P["motion"], P["omega"], P["trbf_center"], P["trbf_scale"], t, raw=raw, quantize=quantize)
    if fused:
        return rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm, Ks, fx["width"], fx["height"],
                             dynamic=ds, **kw)
    m, q, s, o, c = ds.apply_unfused(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"])
    return rasterization(m, q, s, o, c, vm, Ks, fx["width"], fx["height"], **kw)


def _same_forward(a, b):
    (rc, ra, meta), (rc2, ra2, meta2) = a, b
    for k in ("radii", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets"):
        assert torch.equal(meta[k], meta2[k]), k
    vis = meta["radii"] > 0
    for k in ("means2d", "conics", "depths", "opacities"):
        assert torch.equal(meta[k][vis], meta2[k][vis]), k
    assert torch.equal(rc, rc2) and torch.equal(ra, ra2)
    return vis


@pytest.mark.parametrize("C", [1, 3])
@pytest.mark.parametrize("mode", ["classic", "antialiased"])
def test_fused_slice_is_bit_identical_to_slice_then_render(C, mode):
    fx, raw = _params(6000)
    t = 0.37
    outs, grads = [], []
    for fused in (True, False):
        P = _P(raw)
        # deterministic compositing backward: the whole chain is reproducible bit for bit (operator path)
        rc, ra, meta = _render(P, fx, t, C=C, fused=fused, packed=False, rasterize_mode=mode, deterministic=True)
        w = torch.linspace(0.5, 1.5, rc.numel(), device=rc.device).view_as(rc)
        ((rc * w).sum() + 0.3 * ra.sum()).backward()
        outs.append((rc.detach(), ra.detach(), meta))
        grads.append({k: p.grad.clone() for k, p in P.items()})
    vis = _same_forward(*outs)
    assert 0 < int(vis.sum()) < vis.numel()
    for k in KEYS:  # (the VJP code is shared but contractible on purpose, see gs_common.h GS_FP_STRICT: two kernels fuse it differently)
        assert rel_l2(N(grads[0][k]), N(grads[1][k])) < 1e-6, (k, rel_l2(N(grads[0][k]), N(grads[1][k])))
        assert float(grads[0][k].abs().sum()) > 0, k
    # gaussians no camera saw: exact zeros, in every parameter
    unseen = ~(vis.any(0))
    for k in KEYS:
        assert float(grads[0][k][unseen].abs().max()) == 0.0, k


def test_fused_slice_on_the_step_driver_route():
    """The default route (native step driver, float atomics in the compositing backward): forward bit-identical, gradients within
    the atomics' run-to-run noise; partial requires_grad; a second backward (retain_graph) without the prefilled buffers."""
    fx, raw = _params(6000, seed=2)
    outs, grads = [], []
    for fused in (True, False):
        P = _P(raw)
        P["omega"].requires_grad_(False)
        rc, ra, meta = _render(P, fx, 0.61, fused=fused, packed=False)
        rc.sum().backward(retain_graph=True)
        g1 = {k: p.grad.clone() for k, p in P.items() if p.grad is not None}
        for p in P.values():
            p.grad = None
        rc.sum().backward()
        g2 = {k: p.grad.clone() for k, p in P.items() if p.grad is not None}
        assert "omega" not in g1
        for k in g1:
            assert rel_l2(N(g2[k]), N(g1[k])) < 1e-4, k
        outs.append((rc.detach(), ra.detach(), meta))
        grads.append(g1)
    _same_forward(*outs)
    for k in grads[0]:
        assert rel_l2(N(grads[0][k]), N(grads[1][k])) < 1e-4, (k, rel_l2(N(grads[0][k]), N(grads[1][k])))


@pytest.mark.parametrize("C", [1, 2])
def test_fused_raw_parameters_and_round_quantizer_in_the_kernel(C):
    """The trainer's RAW parameters handed over as they are: round STE hooks (in-place clamp) -> exp / sigmoid -> slice in the projection
    kernel, against STE(activation) -> torch.exp(trbf_scale) -> temporal_slice -> rasterization."""
    fx, raw = _params(6000, seed=3, activated=False)
    raw["scales"][:7, 0] = 2.5      # outside [-10, 2]: clamped IN the parameter
    raw["quats"][:5, 1] = -1.75
    raw["opacities"][:9] = 8.0
    raw["colors"][:4, 2] = 7.75
    names = ("scales", "opacities", "trbf_scale")
    outs, grads, Ps = [], [], []
    for fused in (True, False):
        P = _P(raw)
        rc, ra, meta = _render(P, fx, 0.45, C=C, fused=fused, raw=names, quantize=BDS, packed=False, deterministic=True)
        (rc * torch.linspace(0.2, 1.0, rc.numel(), device=rc.device).view_as(rc)).sum().backward()
        outs.append((rc.detach(), ra.detach(), meta))
        grads.append({k: p.grad.clone() for k, p in P.items()})
        Ps.append(P)
    for k in KEYS:  # the same parameters after the call: clamped where out of range, untouched elsewhere
        assert torch.equal(Ps[0][k].detach(), Ps[1][k].detach()), k
    assert float(Ps[0]["scales"][:7, 0].detach().max()) == 2.0 and float(Ps[0]["quats"][:5, 1].detach().min()) == -1.0
    assert float(Ps[0]["opacities"][:9].detach().max()) == 7.0 and float(Ps[0]["colors"][:4, 2].detach().max()) == 7.5
    untouched = np.ones(raw["scales"].shape, bool)
    untouched[:7, 0] = False
    assert np.array_equal(N(Ps[0]["scales"])[untouched], np.clip(raw["scales"], -10.0, 2.0)[untouched])
    # exp(trbf_scale): torch's exp kernel in the chain, expf in the fused kernel -- the one place the two routes may round differently
    ts_same = torch.equal(torch.exp(Ps[0]["trbf_scale"].detach()), torch.exp(Ps[1]["trbf_scale"].detach()))
    assert ts_same
    (rc, ra, meta), (rc2, ra2, meta2) = outs
    same_bins = all(torch.equal(meta[k], meta2[k]) for k in ("radii", "isect_ids", "flatten_ids"))
    if same_bins:
        assert float((rc - rc2).abs().max()) <= 2e-6
    else:  # (a radius on a rounding boundary: a handful of splats may bin differently)
        assert float((meta["radii"] != meta2["radii"]).float().mean()) < 1e-4
    # (rows that differ in the last bit -- the two routes' exp, and the projection chain as the compiler fused it in each kernel -- move
    # this fixture's ill-conditioned log-scale gradient by a few 1e-4; a splat that bins differently takes its whole gradient with it)
    for k in KEYS:
        assert rel_l2(N(grads[0][k]), N(grads[1][k])) < 5e-3, (k, rel_l2(N(grads[0][k]), N(grads[1][k])))


def test_fused_slice_only_activated_by_the_caller_with_quantized_quats():
    """Mixed use: the caller activates (hooks with entropy models outside), only the quaternions are quantized in the kernel."""
    fx, raw = _params(4000, seed=5)
    raw["quats"][:5, 3] = 1.5
    outs, Ps = [], []
    for fused in (True, False):
        P = _P(raw)
        out = _render(P, fx, 0.2, fused=fused, quantize={"quats": BDS["quats"]}, packed=False)
        out[0].sum().backward()
        outs.append((out[0].detach(), out[1].detach(), out[2]))
        Ps.append(P)
    _same_forward(*outs)
    assert float(Ps[0]["quats"][:5, 3].detach().max()) == 1.0 and torch.equal(Ps[0]["quats"].detach(), Ps[1]["quats"].detach())
    for k in KEYS:
        assert rel_l2(N(Ps[0][k].grad), N(Ps[1][k].grad)) < 1e-4, k


def test_fused_slice_with_nine_feature_channels():
    """The spacetime trainer's feature render: the colours do not ride in the splat rows (wide compositing kernels)."""
    fx, raw = _params(5000, seed=7, channels=9)
    outs, grads = [], []
    for fused in (True, False):
        P = _P(raw)
        rc, ra, meta = _render(P, fx, 0.52, fused=fused, packed=False)
        assert rc.shape[-1] == 9
        rc.sum().backward()
        outs.append((rc.detach(), ra.detach(), meta))
        grads.append({k: p.grad.clone() for k, p in P.items()})
    _same_forward(*outs)
    for k in KEYS:
        assert rel_l2(N(grads[0][k]), N(grads[1][k])) < 1e-4, k


@pytest.mark.parametrize("route", ["packed", "viewmat_grad", "tuple"])
def test_routes_outside_the_fused_kernels_take_the_chain(route):
    from gscodec_studio_amd import rasterization

    fx, raw = _params(3000, seed=9)
    P = _P(raw)
    vm, Ks = T(fx["viewmats"][:1]), T(fx["Ks"][:1])
    if route == "viewmat_grad":
        vm.requires_grad_(True)
    dyn = (P["motion"], P["omega"], P["trbf_center"], P["trbf_scale"], 0.4)
    kw = dict(packed=(route == "packed"))
    rc, ra, meta = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], vm, Ks, fx["width"], fx["height"],
                                 dynamic=dyn, **kw)
    rc.sum().backward()
    if route == "viewmat_grad":
        assert vm.grad is not None and float(vm.grad.abs().sum()) > 0
    P2 = _P(raw)
    rc2, ra2, meta2 = _render(P2, fx, 0.4, fused=False, **kw)
    rc2.sum().backward()
    assert torch.equal(rc, rc2)
    for k in KEYS:
        assert rel_l2(N(P[k].grad), N(P2[k].grad)) < 1e-4, k


def test_operator_level_backward_given_the_same_gradient_rows():
    """gs_projection_rows_dyn_bwd == gs_projection_rows_bwd -> gs_temporal_slice_bwd on the SAME gradient rows (no compositing in
    between): every gradient to 1e-6."""
    from gscodec_studio_amd._wrapper import project_rows
    from gscodec_studio_amd.dynamic import DynamicSlice, temporal_slice

    fx, raw = _params(5000, seed=11)
    vm, Ks = T(fx["viewmats"][:2]), T(fx["Ks"][:2])
    res = []
    for fused in (True, False):
        P = _P(raw)
        ds = DynamicSlice(P["motion"], P["omega"], P["trbf_center"], P["trbf_scale"], 0.33)
        if fused:
            out = project_rows(P["means"], None, P["quats"], P["scales"], vm, Ks, fx["width"], fx["height"], P["opacities"], P["colors"],
                               dynamic=ds)
        else:
            m, q, o, _ = temporal_slice(P["means"], P["motion"], P["quats"], P["omega"], P["opacities"], P["trbf_center"], P["trbf_scale"], 0.33)
            out = project_rows(m, None, q, P["scales"], vm, Ks, fx["width"], fx["height"], o, P["colors"])
        radii, means2d, depths, conics, opac, colors, rows = out
        g = torch.Generator(device="cuda:0").manual_seed(5)
        G = torch.randn(rows.shape, device=rows.device, generator=g) * (radii > 0)[..., None]
        torch.autograd.backward([means2d, conics, opac, colors, depths],
                                [G[..., 0:2], G[..., 2:5], G[..., 5], G[..., 6:9], G[..., 9] * (radii > 0)])
        res.append((radii, rows, {k: p.grad.clone() for k, p in P.items()}))
    assert torch.equal(res[0][0], res[1][0])
    vis = res[0][0] > 0
    assert torch.equal(res[0][1][vis][:, :12], res[1][1][vis][:, :12])
    for k in KEYS:
        assert rel_l2(N(res[0][2][k]), N(res[1][2][k])) < 1e-6, (k, rel_l2(N(res[0][2][k]), N(res[1][2][k])))


def test_render_dynamic_matches_the_trainer_call_pattern():
    """``render_dynamic`` (raw parameter dict + STGCompressionSimulation in, image out) against the trainer's own sequence: hooks ->
    torch.exp / sigmoid -> temporal_slice -> rasterization (simple_trainer_dyngs.py:463-554), RGB and the nine STG channels."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd.compression_simulation import STGCompressionSimulation
    from gscodec_studio_amd.dynamic import render_dynamic, temporal_slice

    fx, raw = _params(4000, seed=13, activated=False)
    rs = np.random.RandomState(3)
    raw["features_dir"], raw["features_time"] = rs.randn(4000, 3).astype(np.float32)[:raw["means"].shape[0]], rs.randn(4000, 3).astype(np.float32)[:raw["means"].shape[0]]
    raw["scales"][:3, 1] = 3.0
    vm, Ks, W, H = T(fx["viewmats"][:1]), T(fx["Ks"][:1]), fx["width"], fx["height"]
    for feats in ("colors", "stg"):
        t = 0.42
        sim = STGCompressionSimulation(quantization_sim_type="round", entropy_steps={})
        P = {k: torch.nn.Parameter(T(v)) for k, v in raw.items()}
        rc, ra, info = render_dynamic(P, t, vm, Ks, W, H, compression_sim=sim, step=0, features=feats, packed=False)
        rc.sum().backward()
        P2 = {k: torch.nn.Parameter(T(v)) for k, v in raw.items()}
        q, _ = sim.simulate_compression(P2, step=0)
        scales, opac, tscale = torch.exp(q["scales"]), torch.sigmoid(q["opacities"]), torch.exp(q["trbf_scale"])
        m_t, q_t, o_t, _ = temporal_slice(q["means"], q["motion"], q["quats"], q["omega"], opac, q["trbf_center"], tscale, t)
        cols = q["colors"] if feats == "colors" else torch.cat((q["colors"], q["features_dir"], (t - q["trbf_center"]).detach() * q["features_time"]), 1)
        rc2, ra2, _ = rasterization(m_t, q_t, scales, o_t, cols, vm, Ks, W, H, packed=False)
        rc2.sum().backward()
        assert float(P["scales"][:3, 1].detach().max()) == 2.0
        for k in P:
            assert torch.equal(P[k].detach(), P2[k].detach()), k
        # (torch's exp / sigmoid kernels in the chain, expf in the fused kernel: a splat on a radius / alpha threshold may flip)
        assert_close(N(rc), N(rc2), 1e-4, 1e-5, f"render_dynamic {feats}", max_bad_frac=3e-4)
        for k in P:
            if P2[k].grad is None:
                assert P[k].grad is None or float(P[k].grad.abs().max()) == 0.0, k
                continue
            # (float atomics in the compositing backward + torch's exp / sigmoid in the chain: the ill-conditioned log-scale gradient of
            # this fixture moves by a few 1e-4 from run to run; bit-identity is the business of the tests above)
            assert rel_l2(N(P[k].grad), N(P2[k].grad)) < 2e-3, (feats, k, rel_l2(N(P[k].grad), N(P2[k].grad)))


@pytest.mark.parametrize("raw_mode", [False, True])
def test_temporal_visibility_mask_culls_like_the_trainer_filter(raw_mode):
    """``temp_vis_mask`` of the dynamic trainer (simple_trainer_dyngs.py:526-571): the reference FILTERS the splats whose temporal basis is
    <= 0.05 out of every array before rasterization and re-expands the per-gaussian info afterwards; the fused route culls them in the
    projection.  Same image bit for bit (the same splats reach the same lists in the same order), same gradients (exact zeros for the
    masked splats), full-size info with the reference's t_vis_mask."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd.dynamic import DynamicSlice, render_dynamic, temporal_slice

    fx, raw = _params(6000, seed=21, activated=not raw_mode)
    raw["trbf_scale"] = raw["trbf_scale"] * (0.35 if not raw_mode else 1.0) - (1.0 if raw_mode else 0.0)  # narrow lifetimes: many splats are off at t
    t = 0.3
    vm, Ks, W, H = T(fx["viewmats"][:2]), T(fx["Ks"][:2]), fx["width"], fx["height"]
    # ---- the trainer's way: slice, mask, filter, render, scatter the info back
    P = _P(raw)
    scales, opac, tscale = (torch.exp(P["scales"]), torch.sigmoid(P["opacities"]), torch.exp(P["trbf_scale"])) if raw_mode else \
        (P["scales"], P["opacities"], P["trbf_scale"])
    m_t, q_t, o_t, mask = temporal_slice(P["means"], P["motion"], P["quats"], P["omega"], opac, P["trbf_center"], tscale, t, temp_vis_mask=True)
    assert 0.05 < float(mask.float().mean()) < 0.95, float(mask.float().mean())
    rc0, ra0, info0 = rasterization(m_t[mask], q_t[mask], scales[mask], o_t[mask], P["colors"][mask], vm, Ks, W, H, packed=False)
    rc0.sum().backward()
    # ---- fused
    P2 = _P(raw)
    if raw_mode:
        rc1, ra1, info1 = render_dynamic(P2, t, vm, Ks, W, H, temp_vis_mask=True, packed=False)
        tmask = info1["t_vis_mask"]
    else:
        ds = DynamicSlice(P2["motion"], P2["omega"], P2["trbf_center"], P2["trbf_scale"], t, min_trbf=0.05)
        rc1, ra1, info1 = rasterization(P2["means"], P2["quats"], P2["scales"], P2["opacities"], P2["colors"], vm, Ks, W, H, packed=False, dynamic=ds)
        tmask = ds.t_vis_mask
    rc1.sum().backward()
    if not raw_mode:
        assert torch.equal(tmask, mask)
        assert torch.equal(rc1, rc0) and torch.equal(ra1, ra0)
        assert torch.equal(info1["radii"][:, mask], info0["radii"])
    else:  # (torch's exp / sigmoid in the chain: a basis value on the threshold may fall the other way)
        assert float((tmask != mask).float().mean()) < 1e-3
        assert_close(N(rc1), N(rc0), 1e-4, 1e-5, "temp_vis_mask render", max_bad_frac=3e-4)
    assert int((info1["radii"][:, ~tmask] != 0).sum()) == 0 and info1["radii"].shape == (2, raw["means"].shape[0])
    assert info1["flatten_ids"].numel() == info0["flatten_ids"].numel() or raw_mode
    for k in KEYS:
        g0, g1 = P[k].grad, P2[k].grad
        assert float(g1[~tmask].abs().max()) == 0.0, k       # masked at this timestamp: no gradient at all
        assert rel_l2(N(g1), N(g0)) < (1e-4 if not raw_mode else 5e-3), (k, rel_l2(N(g1), N(g0)))
    # the stand-alone route (packed): masked splats composite nothing
    P3 = _P(raw)
    if not raw_mode:
        ds3 = DynamicSlice(P3["motion"], P3["omega"], P3["trbf_center"], P3["trbf_scale"], t, min_trbf=0.05)
        rc3, _, _ = rasterization(P3["means"], P3["quats"], P3["scales"], P3["opacities"], P3["colors"], vm, Ks, W, H, packed=True, dynamic=ds3)
        assert torch.equal(rc3, rc0) and torch.equal(ds3.t_vis_mask, mask)


@pytest.mark.parametrize("seed", range(12))
def test_fused_slice_fuzz_over_the_call_options(seed):
    """Random scenes, cameras and options (camera model, antialiasing, tile size, clipping planes, radius clip, backgrounds, render
    mode, absgrad, channel count, in-kernel quantizers, temporal culling): the fused route against slice-then-render -- the options
    the hand-written cases above do not enumerate.  Forward bit for bit, backward within the float atomics' noise."""
    import os

    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd.dynamic import DynamicSlice

    rs = np.random.RandomState(9100 + seed + int(os.environ.get("GS_FUZZ_SEED_OFFSET", "0")))
    C, n = int(rs.randint(1, 4)), int(rs.choice([300, 2000, 6000]))
    W, H = int(rs.randint(40, 300)), int(rs.randint(40, 220))
    cm = str(rs.choice(["pinhole", "pinhole", "ortho", "fisheye"]))
    aa = bool(rs.rand() < 0.4)
    ts = int(rs.choice([8, 16]))
    D = int(rs.choice([1, 3, 3, 4, 9]))
    mode = str(rs.choice(["RGB", "RGB", "RGB+D", "RGB+ED", "D"]))
    raw = dict(
        means=(rs.randn(n, 3) * np.array([1.5, 1.5, 1.0])).astype(np.float32), quats=rs.randn(n, 4).astype(np.float32),
        scales=np.exp(rs.uniform(np.log(0.01), np.log(0.25), size=(n, 3))).astype(np.float32), opacities=rs.rand(n).astype(np.float32),
        colors=rs.rand(n, D).astype(np.float32), trbf_center=rs.uniform(0, 1, (n, 1)).astype(np.float32),
        trbf_scale=np.exp(rs.uniform(-1.5, 0.5, (n, 1))).astype(np.float32), motion=(0.1 * rs.randn(n, 9)).astype(np.float32),
        omega=(0.2 * rs.randn(n, 4)).astype(np.float32))
    viewmats = np.tile(np.eye(4, dtype=np.float32), (C, 1, 1))
    for c in range(C):
        a = rs.uniform(-0.5, 0.5)
        viewmats[c, :3, :3] = np.array([[np.cos(a), 0, np.sin(a)], [0, 1, 0], [-np.sin(a), 0, np.cos(a)]], np.float32)
        viewmats[c, :3, 3] = np.array([rs.uniform(-0.3, 0.3), rs.uniform(-0.3, 0.3), rs.uniform(3.0, 5.0)], np.float32)
    f = (0.9 * W) if cm != "ortho" else 40.0
    Ks = np.tile(np.array([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1]], np.float32), (C, 1, 1))
    near, far = (0.01, 1e10) if rs.rand() < 0.6 else (3.2, 6.0)
    kw = dict(near_plane=near, far_plane=far, radius_clip=0.0 if rs.rand() < 0.7 else 2.0, tile_size=ts, camera_model=cm,
              rasterize_mode="antialiased" if aa else "classic", render_mode=mode, packed=False, absgrad=bool(rs.rand() < 0.3),
              backgrounds=T(rs.rand(C, D).astype(np.float32)) if (rs.rand() < 0.5 and mode == "RGB") else None)
    # (the caller activates here -- bit-identical forwards --, so only the hooks that work on activated values: quats, colours; the raw
    # forms go through torch's exp in the chain and expf in the kernel, test_fused_raw_parameters_and_round_quantizer_in_the_kernel)
    q = {"quats": (-1.0, 1.0, 8)} if rs.rand() < 0.4 else {}
    if rs.rand() < 0.4:
        q["colors"] = (0.1, 0.8, 6)
    min_trbf = 0.05 if rs.rand() < 0.3 else None
    t = float(rs.uniform(0, 1))
    tag = f"seed {seed}: C={C} n={n} {W}x{H} {cm} aa={aa} tile {ts} D={D} {mode} q={sorted(q)} min_trbf={min_trbf} absgrad={kw['absgrad']}"
    outs, grads, Ps = [], [], []
    for fused in (True, False):
        P = _P(raw)
        ds = DynamicSlice(P["motion"], P["omega"], P["trbf_center"], P["trbf_scale"], t, quantize=q or None, min_trbf=min_trbf)
        if fused:
            out = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"], T(viewmats), T(Ks), W, H, dynamic=ds, **kw)
        else:
            out = rasterization(*ds.apply_unfused(P["means"], P["quats"], P["scales"], P["opacities"], P["colors"]), T(viewmats), T(Ks), W, H, **kw)
        wgt = torch.linspace(0.5, 1.5, out[0].numel(), device=out[0].device).view_as(out[0])
        ((out[0] * wgt).sum() + 0.3 * out[1].sum()).backward()
        outs.append((out[0].detach(), out[1].detach(), out[2]))
        grads.append({k: (p.grad.clone() if p.grad is not None else None) for k, p in P.items()})
        Ps.append(P)
    if min_trbf is None:
        _same_forward(*outs)
    else:
        # the chain cannot drop splats (it zeroes their opacity: they stay in the lists and composite nothing), the kernel culls them:
        # same image, same radii for the splats that are on at t
        (rc, ra, meta), (rc2, ra2, meta2) = outs
        assert torch.equal(rc, rc2) and torch.equal(ra, ra2), tag
        on = ds.t_vis_mask
        assert torch.equal(meta["radii"][:, on], meta2["radii"][:, on]), tag
        if not ("colors" in q and D != 3):  # (colour hooks on other widths than the rows' three take the chain inside rasterization())
            assert int((meta["radii"][:, ~on] != 0).sum()) == 0, tag
    for k in KEYS:
        assert torch.equal(Ps[0][k].detach(), Ps[1][k].detach()), (k, tag)  # the in-place clamps
        g0, g1 = grads[0][k], grads[1][k]
        assert (g0 is None) == (g1 is None), (k, tag)
        if g0 is None or float(g1.abs().max()) == 0.0:
            assert g0 is None or float(g0.abs().max()) == 0.0, (k, tag)
            continue
        assert rel_l2(N(g0), N(g1)) < 2e-4, (k, tag, rel_l2(N(g0), N(g1)))
    if kw["absgrad"]:
        assert "means2d" in outs[0][2]


@pytest.mark.parametrize("n", [5003, 1024, 1, 257])
@pytest.mark.parametrize("hooks", [(), ("colors", "features_dir", "features_time"), ("features_time",)])
def test_stg_features_equal_cat_and_the_hooks_bit_for_bit(hooks, n):
    """``dynamic.stg_features`` (gs_stg_features_fwd / _bwd): the spacetime trainer's nine colour channels
    cat(colors, features_dir, (t - trbf_center).detach() * features_time) (examples/simple_trainer_STG.py:506-551), optionally with the
    round STE hook of each part in front -- the same values, the same in-place clamps, the same gradients as the torch chain."""
    from gscodec_studio_amd.compression_simulation.ops import STE
    from gscodec_studio_amd.dynamic import stg_features

    rs = np.random.RandomState(11)
    bds = {"colors": (-7.5, 7.5, 8), "features_dir": (-10.0, 10.0, 8), "features_time": (-10.0, 10.0, 8)}
    raw = {k: (rs.randn(n, 3) * 6).astype(np.float32) for k in bds}  # (some values beyond the ranges: clamped in the parameter)
    center = rs.rand(n, 1).astype(np.float32)
    t = 0.37
    v = T(rs.randn(n, 9).astype(np.float32))
    # fused
    P = {k: torch.nn.Parameter(T(x)) for k, x in raw.items()}
    c = torch.nn.Parameter(T(center))
    out = stg_features(P["colors"], P["features_dir"], P["features_time"], c, t, quantize={k: bds[k] for k in hooks})
    (out * v).sum().backward()
    # the chain
    P2 = {k: torch.nn.Parameter(T(x)) for k, x in raw.items()}
    c2 = torch.nn.Parameter(T(center))
    q = {k: (STE.apply(P2[k], bds[k][2], bds[k][0], bds[k][1], 0) if k in hooks else P2[k]) for k in bds}
    ref = torch.cat((q["colors"], q["features_dir"], (t - c2).detach() * q["features_time"]), dim=1)
    (ref * v).sum().backward()
    assert out.shape == (n, 9) and torch.equal(out, ref)
    for k in bds:
        assert torch.equal(P[k].detach(), P2[k].detach()), k  # clamped in place where hooked, untouched elsewhere
        assert torch.equal(P[k].grad, P2[k].grad), k
        if k in hooks:
            assert float(P[k].detach().abs().max()) <= max(abs(bds[k][0]), abs(bds[k][1]))
        else:
            assert np.array_equal(N(P[k]), raw[k])
    assert c.grad is None and c2.grad is None  # tforpoly is detached
    # partial requires_grad
    P3 = {k: T(x).requires_grad_(k == "features_dir") for k, x in raw.items()}
    out3 = stg_features(P3["colors"], P3["features_dir"], P3["features_time"], T(center), t)
    (out3 * v).sum().backward()
    assert P3["colors"].grad is None and P3["features_time"].grad is None and torch.equal(P3["features_dir"].grad, v[:, 3:6])
    # views that are not 16-byte aligned take the element-per-lane kernels: same results
    big = {k: torch.zeros(n * 3 + 1, device="cuda") for k in bds}
    P4 = {}
    for k in bds:
        big[k][1:] = T(raw[k]).reshape(-1)
        P4[k] = big[k][1:].view(n, 3).requires_grad_(True)
    out4 = stg_features(P4["colors"], P4["features_dir"], P4["features_time"], T(center), t, quantize={k: bds[k] for k in hooks})
    assert torch.equal(out4, ref)
    for k in bds:
        assert torch.equal(P4[k].detach(), P2[k].detach()), k
