"""BASELINE configs 3 and 5 as COMPOSITIONS, on the GPU through the product API:

config 3  (examples/simple_trainer.py:906-907 + 742-803 of the reference): CompressionSimulation hooks on the trainer's raw
          parameters -> exp / sigmoid / cat(sh0, shN) -> rasterization(sh_degree=3) -> backward.
          * fixture size: against the hand-chained ORACLE stage VJPs (quantizer -> activations -> projection / SH ->
            binning -> compositing), noise reproduced from the device generator's seed;
          * 1 M splats (the bench scene): size-independent properties.
config 5  (examples/simple_trainer_dyngs.py:463-577, 616-617): STGCompressionSimulation("round") on the 17-float attribute
          set -> activations -> temporal_slice at a timestamp -> rasterization(colors [N,3]) -> backward.
          * fixture size: against the oracle chain (float64 temporal slice + oracle renderer);
          * 2 M splats: properties (parameters clamped in place, finite gradients, exact zeros for culled splats,
            determinism of the integer stages)."""
import numpy as np
import pytest
import torch

from util import N, T, assert_close, garden, garden_sh, rel_l2

pytestmark = pytest.mark.gpu

from oracle import gs_oracle as O  # noqa: E402
from oracle import unfused_oracle as UO  # noqa: E402

BDS3 = dict(scales=(-10.0, 2.0), quats=(-1.0, 1.0), opacities=(-15.0, 15.0), sh0=(-2.0, 4.0))


def _raw_params(n, scale_mult=5.0):
    """The trainer's raw parameters for the garden fixture: log-scales, opacity logits, sh0 / shN."""
    fx = garden(n, scale_mult=scale_mult)
    sh = garden_sh(fx["rgb"], K=16)
    raw = dict(means=fx["means"], scales=np.log(fx["scales"]).astype(np.float32), quats=fx["quats"],
               opacities=np.log(fx["opacities"] / (1 - fx["opacities"])).astype(np.float32),
               sh0=np.ascontiguousarray(sh[:, :1]), shN=np.ascontiguousarray(sh[:, 1:]))
    # a few out-of-range values so the noise quantizer's gradient mask is exercised
    raw["scales"][:7, 0] = -11.5
    raw["sh0"][:5, 0, 1] = 4.5
    return fx, raw


def test_config3_hooks_render_backward_vs_oracle_chain():
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd.compression_simulation import CompressionSimulation

    n, cams = 2500, 2
    fx, raw = _raw_params(n)
    W, H = fx["width"], fx["height"]
    P = {k: torch.nn.Parameter(T(v)) for k, v in raw.items()}
    sim = CompressionSimulation(entropy_model_enable=False, entropy_steps={})
    torch.manual_seed(1234)
    q, bits = sim.simulate_compression(P, step=0)
    assert all(b is None for b in bits.values())
    # the same noise, drawn again from the same seed in the hooks' order (scales, quats, opacities, sh0)
    torch.manual_seed(1234)
    noise = {k: N(torch.empty_like(P[k]).uniform_(-0.5, 0.5)) for k in ("scales", "quats", "opacities", "sh0")}
    o_q = {k: O.quant_noise_fwd(raw[k], noise[k], lo, hi, (hi - lo) / 255) for k, (lo, hi) in BDS3.items()}
    for k in BDS3:
        assert np.array_equal(N(q[k]), o_q[k]), k  # bit-exact quantizer inside the composition
    assert q["shN"] is P["shN"] and torch.equal(q["means"], P["means"])

    scales, opac = torch.exp(q["scales"]), torch.sigmoid(q["opacities"])
    sh = torch.cat([q["sh0"], q["shN"]], dim=1)
    vm, Ks = T(fx["viewmats"][:cams]), T(fx["Ks"][:cams])
    rc, ra, meta = rasterization(q["means"], q["quats"], scales, opac, sh, vm, Ks, W, H, sh_degree=3, packed=False)

    # ---- oracle forward on the SAME quantized values (fp32 exp / sigmoid on the host)
    o_scales = np.exp(o_q["scales"]).astype(np.float32)
    o_opac = (1.0 / (1.0 + np.exp(-o_q["opacities"].astype(np.float64)))).astype(np.float32)
    o_sh = np.concatenate([o_q["sh0"], raw["shN"]], 1)
    o_rc, o_ra, om = O.rasterization(raw["means"], o_q["quats"], o_scales, o_opac, o_sh, fx["viewmats"][:cams], fx["Ks"][:cams], W, H,
                                     sh_degree=3)
    _, _, _, bl = O.rasterize_fwd(om["means2d"], om["conics"], om["colors"], om["opacities"], W, H, 16, om["isect_offsets"],
                                  om["flatten_ids"], return_borderline=True)
    ok = bl == 0
    assert_close(N(rc)[ok], o_rc[ok], 1e-4, 5e-5, "config 3 render", max_bad_frac=2e-4)
    assert_close(N(ra)[ok], o_ra[ok], 1e-4, 5e-5, "config 3 alpha", max_bad_frac=2e-4)

    rs = np.random.RandomState(7)
    v_rc = rs.randn(*o_rc.shape).astype(np.float32) * ok[..., None]
    ((rc * T(v_rc)).sum()).backward()

    # ---- oracle backward chain
    C = cams
    v_m2, v_cn, v_col, v_op, _ = O.rasterize_bwd(om["means2d"], om["conics"], om["colors"], om["opacities"], W, H, 16,
                                                 om["isect_offsets"], om["flatten_ids"], o_ra, om["last_ids"], v_rc, np.zeros_like(o_ra))
    c2w = np.linalg.inv(fx["viewmats"][:cams].astype(np.float64)).astype(np.float32)
    dirs = raw["means"][None] - c2w[:, None, :3, 3]
    shs = np.ascontiguousarray(np.broadcast_to(o_sh[None], (C,) + o_sh.shape))
    sh_raw = O.sh_fwd(3, dirs, shs, om["radii"] > 0)
    v_coeffs, v_dirs = O.sh_bwd(3, dirs, shs, v_col * ((sh_raw + 0.5) > 0), om["radii"] > 0)
    g_means, _, g_quats, g_scales_act, _ = O.projection_bwd(raw["means"], None, o_q["quats"], o_scales, fx["viewmats"][:cams],
                                                             fx["Ks"][:cams], W, H, 0.3, "pinhole", om["radii"], om["conics"], None,
                                                             v_m2, np.zeros_like(om["depths"]), v_cn, None, need_viewmats=False)
    g_sh = v_coeffs.sum(0)
    expect = dict(
        means=g_means + v_dirs.sum(0),
        quats=O.quant_noise_bwd(raw["quats"], g_quats, *BDS3["quats"]),
        scales=O.quant_noise_bwd(raw["scales"], g_scales_act * o_scales, *BDS3["scales"]),                 # d exp
        opacities=O.quant_noise_bwd(raw["opacities"], v_op.sum(0) * o_opac * (1 - o_opac), *BDS3["opacities"]),  # d sigmoid
        sh0=O.quant_noise_bwd(raw["sh0"], np.ascontiguousarray(g_sh[:, :1]), *BDS3["sh0"]),
        shN=np.ascontiguousarray(g_sh[:, 1:]),
    )
    for k, ref in expect.items():
        got = N(P[k].grad)
        assert got.shape == ref.shape, k
        assert rel_l2(got, ref) < 5e-4, (k, rel_l2(got, ref))
    # the gradient mask of the noise quantizer: exact zeros outside the bounds
    assert float(P["scales"].grad[:7, 0].abs().max()) == 0.0 and float(P["sh0"].grad[:5, 0, 1].abs().max()) == 0.0
    # noise mode leaves the parameters alone
    for k in raw:
        assert np.array_equal(N(P[k]), raw[k]), k


def test_config3_fused_form_matches_the_reference_call_pattern():
    """The opt-in fused form of the config-3 step -- simulate_compression(activate=True) + rasterization(colors=(sh0, shN)) --
    against the reference's call pattern (hooks, torch.exp / torch.sigmoid / torch.cat, simple_trainer.py:779-800) on the
    same generator state: same image to fp32 rounding of the two activations, same parameter gradients."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd.compression_simulation import CompressionSimulation

    n, cams = 2500, 2
    fx, raw = _raw_params(n)
    W, H = fx["width"], fx["height"]
    vm, Ks = T(fx["viewmats"][:cams]), T(fx["Ks"][:cams])
    sim = CompressionSimulation(entropy_model_enable=False, entropy_steps={})
    rs = np.random.RandomState(3)
    v_rc = T(rs.randn(cams, H, W, 3).astype(np.float32))

    def run(fused):
        P = {k: torch.nn.Parameter(T(v)) for k, v in raw.items()}
        torch.manual_seed(4321)
        if fused:
            q, _ = sim.simulate_compression(P, step=0, activate=True)
            scales, opac, sh = q["scales"], q["opacities"], (q["sh0"], q["shN"])
        else:
            q, _ = sim.simulate_compression(P, step=0)
            scales, opac, sh = torch.exp(q["scales"]), torch.sigmoid(q["opacities"]), torch.cat([q["sh0"], q["shN"]], dim=1)
        rc, ra, _ = rasterization(q["means"], q["quats"], scales, opac, sh, vm, Ks, W, H, sh_degree=3, packed=False)
        (rc * v_rc).sum().backward()
        return N(rc), {k: N(p.grad) for k, p in P.items()}

    rc0, g0 = run(False)
    rc1, g1 = run(True)
    assert_close(rc1, rc0, 1e-5, 1e-6, "fused config-3 render", max_bad_frac=1e-4)
    for k in g0:
        assert rel_l2(g1[k], g0[k]) < 2e-4, (k, rel_l2(g1[k], g0[k]))


def test_config3_full_size_properties():
    """1,006,065 gaussians, SH degree 3, 1080p: hooks -> activations -> render -> backward (bench.py --quantize's step)."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd._helper import sh_workload
    from gscodec_studio_amd.compression_simulation import CompressionSimulation

    w = sh_workload(scene_grid=3, device="cuda:0")
    assert w["N"] == 1006065
    with torch.no_grad():
        raw = dict(means=w["means"].clone(), scales=w["scales"].log(), quats=w["quats"].clone(),
                   opacities=torch.logit(w["opacities"].clamp(1e-6, 1 - 1e-6)), sh0=w["sh"][:, :1].clone(), shN=w["sh"][:, 1:].clone())
    P = {k: torch.nn.Parameter(v.contiguous()) for k, v in raw.items()}
    before = {k: v.detach().clone() for k, v in P.items()}
    sim = CompressionSimulation(entropy_model_enable=False, entropy_steps={})

    def step(seed):
        for p in P.values():
            p.grad = None
        torch.manual_seed(seed)
        q, _ = sim.simulate_compression(P, step=0)
        rc, ra, meta = rasterization(q["means"], q["quats"], torch.exp(q["scales"]), torch.sigmoid(q["opacities"]),
                                     torch.cat([q["sh0"], q["shN"]], 1), w["viewmats"], w["Ks"], w["width"], w["height"],
                                     sh_degree=3, packed=False)
        rc.sum().backward()
        return q, rc, ra, meta

    q, rc, ra, meta = step(5)
    for k, (lo, hi) in BDS3.items():  # |quantized - clamp(param)| <= q_step / 2
        err = (q[k].detach() - before[k].clamp(lo, hi)).abs().max()
        assert float(err) <= (hi - lo) / 255 / 2 * 1.0001, k
        assert torch.equal(P[k].detach(), before[k])
    ra = ra.detach()
    assert bool(torch.isfinite(rc).all()) and float(ra.min()) >= 0 and float(ra.max()) <= 1
    vis = meta["radii"][0] > 0
    n_isects = meta["flatten_ids"].numel()
    assert int(meta["tiles_per_gauss"].sum()) == n_isects and bool((meta["isect_ids"][1:] >= meta["isect_ids"][:-1]).all())
    for k, p in P.items():
        assert p.grad is not None and bool(torch.isfinite(p.grad).all()), k
    # culled splats: exactly zero gradient through every stage and hook
    for k in ("scales", "quats", "opacities", "sh0", "shN"):
        assert float(P[k].grad[~vis].abs().max()) == 0.0, k
    # the same seed gives the same quantized splats, hence bit-identical integer stages and image
    g1 = {k: p.grad.clone() for k, p in P.items()}
    q2, rc2, ra2, meta2 = step(5)
    assert torch.equal(meta2["isect_ids"], meta["isect_ids"]) and torch.equal(meta2["flatten_ids"], meta["flatten_ids"])
    assert torch.equal(rc2, rc)
    for k in P:  # float atomics: run-to-run noise only (the quaternion gradient is the most ill-conditioned: up to 1e-4)
        assert rel_l2(N(P[k].grad), N(g1[k])) < 1e-4, k
    # a different seed changes the noise and therefore the image
    _, rc3, _, _ = step(6)
    assert not torch.equal(rc3, rc)


# ---------------------------------------------------------------------------------------------------------------------
DYN = ("means", "scales", "quats", "opacities", "trbf_center", "trbf_scale", "motion", "omega", "colors", "features_dir", "features_time")
BDS5 = dict(scales=(-10.0, 2.0), quats=(-1.0, 1.0), opacities=(-7.0, 7.0), colors=(-7.5, 7.5), features_dir=(-10.0, 10.0),
            features_time=(-10.0, 10.0))


def _dyn_params(means, scales, quats, opacities, rgb, seed=0):
    n = means.shape[0]
    rs = np.random.RandomState(seed)
    logit = lambda p: np.log(p / (1 - p))  # noqa: E731
    return dict(
        means=means.astype(np.float32), scales=np.log(np.maximum(scales, 1e-6)).astype(np.float32), quats=quats.astype(np.float32),
        opacities=logit(np.clip(opacities, 1e-4, 1 - 1e-4)).astype(np.float32),
        trbf_center=rs.uniform(0, 1, (n, 1)).astype(np.float32), trbf_scale=rs.uniform(-1.5, 0.5, (n, 1)).astype(np.float32),
        motion=(0.02 * rs.randn(n, 9)).astype(np.float32), omega=(0.1 * rs.randn(n, 4)).astype(np.float32),
        colors=rgb.astype(np.float32), features_dir=rs.randn(n, 3).astype(np.float32), features_time=rs.randn(n, 3).astype(np.float32))


def _dyn_step(P, sim, viewmats, Ks, W, H, timestamp):
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd.dynamic import temporal_slice

    q, _ = sim.simulate_compression(P, step=0)
    scales, opac, tscale = torch.exp(q["scales"]), torch.sigmoid(q["opacities"]), torch.exp(q["trbf_scale"])
    means_t, quats_t, opac_t, _ = temporal_slice(q["means"], q["motion"], q["quats"], q["omega"], opac, q["trbf_center"], tscale, timestamp)
    rc, ra, meta = rasterization(means_t, quats_t, scales, opac_t, q["colors"], viewmats, Ks, W, H, packed=False)
    return q, rc, ra, meta


def test_config5_slice_quantize_render_vs_oracle_chain():
    from gscodec_studio_amd.compression_simulation import STGCompressionSimulation

    n, timestamp = 2500, 0.37
    fx = garden(n, scale_mult=5.0)
    W, H = fx["width"], fx["height"]
    raw = _dyn_params(fx["means"], fx["scales"], fx["quats"], fx["opacities"], fx["rgb"])
    raw["colors"][:9, 1] = 9.0  # out of range: "round" clamps the PARAMETER in place and still passes the gradient
    P = {k: torch.nn.Parameter(T(v)) for k, v in raw.items()}
    sim = STGCompressionSimulation(quantization_sim_type="round", entropy_steps={})
    vm, Ks = fx["viewmats"][:1], fx["Ks"][:1]
    q, rc, ra, meta = _dyn_step(P, sim, T(vm), T(Ks), W, H, timestamp)

    # ---- oracle: round quantizer (bit-exact), activations, float64 temporal slice, oracle renderer
    o_q, clamped = {}, {}
    for k, (lo, hi) in BDS5.items():
        clamped[k], o_q[k] = O.quant_round_fwd(raw[k], lo, hi, 8)
        assert np.array_equal(N(q[k]), o_q[k]), k
        assert np.array_equal(N(P[k]), clamped[k]), k  # clamped in place (reference ops.py:63)
    assert float(P["colors"][:9, 1].detach().max()) == 7.5
    o_scales = np.exp(o_q["scales"]).astype(np.float32)
    o_opac = (1.0 / (1.0 + np.exp(-o_q["opacities"].astype(np.float64)))).astype(np.float32)
    o_ts = np.exp(raw["trbf_scale"]).astype(np.float32)
    t64 = lambda a: torch.tensor(a, dtype=torch.float64)  # noqa: E731
    m_t, q_t, o_t, _ = UO.temporal_slice(t64(raw["means"]), t64(raw["motion"]), t64(o_q["quats"]), t64(raw["omega"]), t64(o_opac),
                                         t64(raw["trbf_center"]), t64(o_ts), timestamp)
    m_t, q_t, o_t = (x.numpy().astype(np.float32) for x in (m_t, q_t, o_t))
    o_rc, o_ra, om = O.rasterization(m_t, q_t, o_scales, o_t.reshape(-1), o_q["colors"], vm, Ks, W, H)
    _, _, _, bl = O.rasterize_fwd(om["means2d"], om["conics"], om["colors"], om["opacities"], W, H, 16, om["isect_offsets"],
                                  om["flatten_ids"], return_borderline=True)
    ok = bl == 0
    assert_close(N(rc)[ok], o_rc[ok], 1e-4, 5e-5, "config 5 render", max_bad_frac=3e-4)

    rs = np.random.RandomState(3)
    v_rc = rs.randn(*o_rc.shape).astype(np.float32) * ok[..., None]
    (rc * T(v_rc)).sum().backward()
    v_m2, v_cn, v_col, v_op, _ = O.rasterize_bwd(om["means2d"], om["conics"], om["colors"], om["opacities"], W, H, 16,
                                                 om["isect_offsets"], om["flatten_ids"], o_ra, om["last_ids"], v_rc, np.zeros_like(o_ra))
    g_mt, _, g_qt, g_sc, _ = O.projection_bwd(m_t, None, q_t, o_scales, vm, Ks, W, H, 0.3, "pinhole", om["radii"], om["conics"], None, v_m2,
                                              np.zeros_like(om["depths"]), v_cn, None, need_viewmats=False)
    keys = ["means", "motion", "quats", "omega", "opacities", "trbf_center", "trbf_scale"]
    ins = [raw["means"], raw["motion"], o_q["quats"], raw["omega"], o_opac, raw["trbf_center"], o_ts]
    _, grads = UO.with_grads(lambda *a: UO.temporal_slice(*a, timestamp)[:3], ins, (g_mt, g_qt, v_op[0]))
    gs = dict(zip(keys, grads))
    expect = dict(
        means=gs["means"], motion=gs["motion"], omega=gs["omega"], quats=gs["quats"],                      # round STE: identity
        opacities=gs["opacities"].reshape(-1) * o_opac * (1 - o_opac), scales=g_sc * o_scales,
        trbf_center=gs["trbf_center"].reshape(n, 1), trbf_scale=gs["trbf_scale"].reshape(n, 1) * o_ts, colors=v_col[0])
    for k, ref in expect.items():
        got = N(P[k].grad)
        assert rel_l2(got, ref.reshape(got.shape)) < 1e-3, (k, rel_l2(got, ref.reshape(got.shape)))
    # features_dir / features_time are quantized by the hooks but do not reach the renderer in this trainer (dyngs.py:519-520)
    assert P["features_dir"].grad is None or float(P["features_dir"].grad.abs().max()) == 0.0


def test_spacetime_trainer_nine_channel_feature_render_vs_oracle_chain():
    """The STG trainer's step (examples/simple_trainer_STG.py:506-551): hooks (round) -> temporal slice -> colors_precomp =
    cat(feature_color, feature_dir, tforpoly * feature_time) -> a 9-CHANNEL render -> backward.  This is the one consumer of the
    features_dir / features_time tensors the STG hooks quantize; the render runs on the wide compositing kernels (round 5).
    Forward against the oracle's pipeline, every parameter gradient against the hand-chained oracle VJPs."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd.compression_simulation import STGCompressionSimulation
    from gscodec_studio_amd.dynamic import temporal_slice

    n, timestamp = 2500, 0.41
    fx = garden(n, scale_mult=5.0)
    W, H = fx["width"], fx["height"]
    raw = _dyn_params(fx["means"], fx["scales"], fx["quats"], fx["opacities"], fx["rgb"], seed=4)
    P = {k: torch.nn.Parameter(T(v)) for k, v in raw.items()}
    sim = STGCompressionSimulation(quantization_sim_type="round", entropy_steps={})
    vm, Ks = fx["viewmats"][:1], fx["Ks"][:1]
    q, _ = sim.simulate_compression(P, step=0)
    scales, opac, tscale = torch.exp(q["scales"]), torch.sigmoid(q["opacities"]), torch.exp(q["trbf_scale"])
    means_t, quats_t, opac_t, _ = temporal_slice(q["means"], q["motion"], q["quats"], q["omega"], opac, q["trbf_center"], tscale, timestamp)
    tforpoly = (timestamp - q["trbf_center"]).detach()                                              # STG.py:506-509, 524
    feats = torch.cat((q["colors"], q["features_dir"], tforpoly * q["features_time"]), dim=1)       # STG.py:531
    rc, ra, meta = rasterization(means_t, quats_t, scales, opac_t, feats, T(vm), T(Ks), W, H, packed=False)
    assert rc.shape[-1] == 9

    o_q = {k: O.quant_round_fwd(raw[k], lo, hi, 8)[1] for k, (lo, hi) in BDS5.items()}
    for k in BDS5:
        assert np.array_equal(N(q[k]), o_q[k]), k
    o_scales = np.exp(o_q["scales"]).astype(np.float32)
    o_opac = (1.0 / (1.0 + np.exp(-o_q["opacities"].astype(np.float64)))).astype(np.float32)
    o_ts = np.exp(raw["trbf_scale"]).astype(np.float32)
    t64 = lambda a: torch.tensor(a, dtype=torch.float64)  # noqa: E731
    m_t, q_t, o_t, _ = UO.temporal_slice(t64(raw["means"]), t64(raw["motion"]), t64(o_q["quats"]), t64(raw["omega"]), t64(o_opac),
                                         t64(raw["trbf_center"]), t64(o_ts), timestamp)
    m_t, q_t, o_t = (x.numpy().astype(np.float32) for x in (m_t, q_t, o_t))
    tfp = (np.float32(timestamp) - raw["trbf_center"]).astype(np.float32)
    o_feats = np.concatenate([o_q["colors"], o_q["features_dir"], tfp * o_q["features_time"]], 1).astype(np.float32)
    assert np.allclose(N(feats), o_feats, rtol=0, atol=1e-6)
    _, _, om = O.rasterization(m_t, q_t, o_scales, o_t.reshape(-1), o_feats, vm, Ks, W, H)
    cols = o_feats[None]
    o_rc, o_ra, o_li, bl = O.rasterize_fwd(om["means2d"], om["conics"], cols, om["opacities"], W, H, 16, om["isect_offsets"], om["flatten_ids"],
                                           return_borderline=True)
    ok = bl == 0
    assert_close(N(rc)[ok], o_rc[ok], 1e-4, 5e-5, "9-channel feature render", max_bad_frac=3e-4)

    rs = np.random.RandomState(8)
    v_rc = rs.randn(*o_rc.shape).astype(np.float32) * ok[..., None]
    (rc * T(v_rc)).sum().backward()
    v_m2, v_cn, v_col, v_op, _ = O.rasterize_bwd(om["means2d"], om["conics"], cols, om["opacities"], W, H, 16, om["isect_offsets"],
                                                 om["flatten_ids"], o_ra, o_li, v_rc, np.zeros_like(o_ra))
    g_mt, _, g_qt, g_sc, _ = O.projection_bwd(m_t, None, q_t, o_scales, vm, Ks, W, H, 0.3, "pinhole", om["radii"], om["conics"], None, v_m2,
                                              np.zeros_like(om["depths"]), v_cn, None, need_viewmats=False)
    keys = ["means", "motion", "quats", "omega", "opacities", "trbf_center", "trbf_scale"]
    ins = [raw["means"], raw["motion"], o_q["quats"], raw["omega"], o_opac, raw["trbf_center"], o_ts]
    _, grads = UO.with_grads(lambda *a: UO.temporal_slice(*a, timestamp)[:3], ins, (g_mt, g_qt, v_op[0]))
    gs = dict(zip(keys, grads))
    expect = dict(
        means=gs["means"], motion=gs["motion"], omega=gs["omega"], quats=gs["quats"],
        opacities=gs["opacities"].reshape(-1) * o_opac * (1 - o_opac), scales=g_sc * o_scales,
        trbf_center=gs["trbf_center"].reshape(n, 1), trbf_scale=gs["trbf_scale"].reshape(n, 1) * o_ts,
        colors=v_col[0][:, 0:3], features_dir=v_col[0][:, 3:6], features_time=tfp * v_col[0][:, 6:9])  # round STE: identity
    for k, ref in expect.items():
        got = N(P[k].grad)
        assert rel_l2(got, ref.reshape(got.shape)) < 1e-3, (k, rel_l2(got, ref.reshape(got.shape)))


def test_config5_full_size_properties():
    """2 M dynamic splats, one 1080p camera: temporal_slice -> round-quantize (17 floats) -> rasterization([N,3]) -> backward."""
    from gscodec_studio_amd._helper import load_test_data
    from gscodec_studio_amd.compression_simulation import STGCompressionSimulation

    d = load_test_data(device="cuda:0", scene_grid=5)  # 2,794,625 gaussians; the first 2 M of a shuffle
    g = torch.Generator(device="cuda:0").manual_seed(0)
    sel = torch.randperm(d[0].shape[0], device="cuda:0", generator=g)[:2_000_000]
    means, quats, scales, opac, rgb, viewmats, Ks, W0, H0 = d
    W, H = 1920, 1080
    Ks = Ks[:1].clone()
    Ks[:, 0] *= W / W0
    Ks[:, 1] *= H / H0
    raw = _dyn_params(N(means[sel]), N(scales[sel]), N(quats[sel]), N(opac[sel]), N(rgb[sel]), seed=1)
    raw["scales"][:11, 2] = 3.0  # outside [-10, 2]
    P = {k: torch.nn.Parameter(T(v)) for k, v in raw.items()}
    sim = STGCompressionSimulation(quantization_sim_type="round", entropy_steps={})
    q, rc, ra, meta = _dyn_step(P, sim, viewmats[:1].contiguous(), Ks, W, H, 0.5)
    rc.sum().backward()
    assert rc.shape == (1, H, W, 3) and bool(torch.isfinite(rc).all())
    assert float(ra.detach().min()) >= 0 and float(ra.detach().max()) <= 1
    # parameters clamped in place; quantized values on the 255-level grid
    assert float(P["scales"][:11, 2].detach().max()) == 2.0
    for k, (lo, hi) in BDS5.items():
        assert float(P[k].detach().min()) >= lo and float(P[k].detach().max()) <= hi, k
        lv = (q[k].detach() - lo) / ((hi - lo) / 255)
        assert float((lv - lv.round()).abs().max()) < 2e-3, k
    vis = meta["radii"][0] > 0
    assert 0 < int(vis.sum()) < 2_000_000
    n_isects = meta["flatten_ids"].numel()
    assert int(meta["tiles_per_gauss"].sum()) == n_isects and bool((meta["isect_ids"][1:] >= meta["isect_ids"][:-1]).all())
    for k in ("means", "scales", "quats", "opacities", "trbf_center", "trbf_scale", "motion", "omega", "colors"):
        assert P[k].grad is not None and bool(torch.isfinite(P[k].grad).all()), k
        assert float(P[k].grad[~vis].abs().max()) == 0.0, k  # culled at this timestamp / by the frustum: exact zeros
    assert float(P["colors"].grad[vis].abs().sum()) > 0
    # second pass from the (now clamped) parameters: identical integer stages and image
    for p in P.values():
        p.grad = None
    q2, rc2, _, meta2 = _dyn_step(P, sim, viewmats[:1].contiguous(), Ks, W, H, 0.5)
    assert torch.equal(meta2["flatten_ids"], meta["flatten_ids"]) and torch.equal(rc2, rc)


def test_config5_full_size_fused_forms_match_the_chain():
    """BASELINE config 5 at its full size (2 M dynamic splats, one 1080p camera): the three fused forms of the frame render against the
    trainer's own sequence (hooks -> exp / sigmoid -> temporal_slice -> rasterization, simple_trainer_dyngs.py:463-554):
    * ``rasterization(dynamic=...)`` on hook outputs: binning and image BIT-IDENTICAL, gradients within the compositing atomics' noise;
    * ``render_dynamic`` (round hooks + activations + slice inside the projection kernels, raw parameters in): the parameters end up
      clamped exactly as the hooks clamp them; image within 1e-4 (torch's exp / sigmoid vs the kernel's), gradients within 2e-3."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd._helper import DYNAMIC_KEYS, dynamic_workload
    from gscodec_studio_amd.compression_simulation import STGCompressionSimulation
    from gscodec_studio_amd.dynamic import render_dynamic, temporal_slice

    w = dynamic_workload(2_000_000, 1920, 1080, device="cuda:0")
    W, H, vm, Ks, t = w["width"], w["height"], w["viewmats"], w["Ks"], 0.5
    w["scales"][:13, 1] = -11.5  # outside [-10, 2]: clamped in the parameter by every form (the low side: a screen-filling splat at the
    #                              upper bound sums millions of cancelling atomics and drowns the comparison in its own noise)

    def params():
        return {k: w[k].clone().requires_grad_(True) for k in DYNAMIC_KEYS}

    def chain(P, fused_slice):
        sim = STGCompressionSimulation(quantization_sim_type="round", entropy_steps={})
        q, _ = sim.simulate_compression(P, step=1)
        scales, opac, tscale = torch.exp(q["scales"]), torch.sigmoid(q["opacities"]), torch.exp(q["trbf_scale"])
        if fused_slice:
            return rasterization(q["means"], q["quats"], scales, opac, q["colors"], vm, Ks, W, H, packed=False,
                                 dynamic=(q["motion"], q["omega"], q["trbf_center"], tscale, t))
        m_t, q_t, o_t, _ = temporal_slice(q["means"], q["motion"], q["quats"], q["omega"], opac, q["trbf_center"], tscale, t)
        return rasterization(m_t, q_t, scales, o_t, q["colors"], vm, Ks, W, H, packed=False)

    res = {}
    for name in ("chain", "slice", "full"):
        P = params()
        if name == "full":
            sim = STGCompressionSimulation(quantization_sim_type="round", entropy_steps={})
            rc, ra, meta = render_dynamic(P, t, vm, Ks, W, H, compression_sim=sim, step=1, packed=False)
        else:
            rc, ra, meta = chain(P, name == "slice")
        rc.sum().backward()
        res[name] = (rc.detach(), meta, {k: p.grad for k, p in P.items() if p.grad is not None}, {k: p.detach() for k, p in P.items()})
    rc0, m0, g0, p0 = res["chain"]
    vis = m0["radii"] > 0
    assert 100_000 < int(vis.sum()) < 2_000_000
    # ---- the slice inside the projection: identical
    rc1, m1, g1, p1 = res["slice"]
    for k in ("radii", "tiles_per_gauss", "isect_ids", "flatten_ids", "isect_offsets"):
        assert torch.equal(m0[k], m1[k]), k
    for k in ("means2d", "conics", "depths", "opacities"):
        assert torch.equal(m0[k][vis], m1[k][vis]), k
    assert torch.equal(rc0, rc1)
    for k in g0:
        assert rel_l2(N(g1[k]), N(g0[k])) < 1e-4, (k, rel_l2(N(g1[k]), N(g0[k])))
    # ---- everything inside the projection
    rc2, m2, g2, p2 = res["full"]
    for k in DYNAMIC_KEYS:
        assert torch.equal(p2[k], p0[k]), k            # same in-place clamps, nothing else touched
    assert float(p2["scales"][:13, 1].min()) == -10.0 and float(p2["scales"][:13, 1].max()) == -10.0
    same = (m2["radii"] == m0["radii"]).float().mean()
    assert float(same) > 0.9999, float(same)
    assert_close(N(rc2), N(rc0), 1e-4, 1e-5, "render_dynamic vs the trainer's sequence", max_bad_frac=1e-4)
    for k in g0:
        if k in ("features_dir", "features_time"):
            continue
        assert rel_l2(N(g2[k]), N(g0[k])) < 2e-3, (k, rel_l2(N(g2[k]), N(g0[k])))
    print(f"[config 5, full size] V = {int(vis.sum())}  I = {m0['flatten_ids'].numel()}  radii equal {float(same) * 100:.4f} %  "
          + "  ".join(f"d/d {k} {rel_l2(N(g2[k]), N(g0[k])):.1e}" for k in ("means", "scales", "quats", "motion", "omega", "trbf_center")))
