"""GPU parity of the fused temporal slicing (gs_temporal_slice_fwd/bwd) against tests/golden/dynamic.npz -- outputs and
autograd gradients of the reference trainer's OWN statements (examples/simple_trainer_dyngs.py:506-536, executed on CPU by
tests/golden/make_golden_dynamic.py) -- and against the float64 restatement of the formulas
(oracle/unfused_oracle.py:temporal_slice, pinned by the same script)."""
import numpy as np
import pytest
import torch

from util import N, T, golden

pytestmark = pytest.mark.gpu

from oracle import unfused_oracle as UO  # noqa: E402


def _inputs(n=5000, seed=0):
    rs = np.random.RandomState(seed)
    return dict(
        means=rs.randn(n, 3).astype(np.float32), motion=(0.3 * rs.randn(n, 9)).astype(np.float32),
        quats=rs.randn(n, 4).astype(np.float32), omega=(0.5 * rs.randn(n, 4)).astype(np.float32),
        opacities=rs.uniform(0.05, 1.0, n).astype(np.float32), trbf_center=rs.uniform(0, 1, (n, 1)).astype(np.float32),
        trbf_scale=np.exp(rs.uniform(-2.5, 0.5, (n, 1))).astype(np.float32))


@pytest.mark.parametrize("timestamp", [0.0, 0.37, 1.0])
def test_temporal_slice_fwd_bwd(timestamp):
    from gscodec_studio_amd.dynamic import temporal_slice

    d = _inputs()
    keys = ["means", "motion", "quats", "omega", "opacities", "trbf_center", "trbf_scale"]
    ins = [T(d[k]).requires_grad_(True) for k in keys]
    m, q, o, mask = temporal_slice(*ins, timestamp, temp_vis_mask=True)
    rs = np.random.RandomState(1)
    vm, vq, vo = (rs.randn(*t.shape).astype(np.float32) for t in (m, q, o))
    ((m * T(vm)).sum() + (q * T(vq)).sum() + (o * T(vo)).sum()).backward()
    outs, grads = UO.with_grads(lambda *a: UO.temporal_slice(*a, timestamp)[:3], [d[k] for k in keys], (vm, vq, vo))
    for got, want, name in zip((m, q, o), outs, ("means_t", "quats_t", "opacity_t")):
        assert np.abs(N(got) - want).max() <= 1e-4 * np.abs(want).max() + 1e-6, name
    tr = UO.temporal_slice(*[torch.tensor(d[k], dtype=torch.float64) for k in keys], timestamp)[3].numpy()
    sure = np.abs(tr - 0.05) > 1e-5
    assert np.array_equal(N(mask)[sure], (tr > 0.05)[sure])
    for p, want, name in zip(ins, grads, keys):
        want = want.reshape(p.shape)
        scale = np.abs(want).max() + 1e-12
        assert np.abs(N(p.grad) - want).max() <= 2e-4 * scale, (name, float(np.abs(N(p.grad) - want).max() / scale))


def test_temporal_slice_selective_grads_and_errors():
    from gscodec_studio_amd.dynamic import temporal_slice

    d = _inputs(n=257)
    means, motion, quats, omega = T(d["means"]), T(d["motion"]).requires_grad_(True), T(d["quats"]), T(d["omega"])
    opac, c, s = T(d["opacities"]).requires_grad_(True), T(d["trbf_center"]), T(d["trbf_scale"]).requires_grad_(True)
    m, q, o, mask = temporal_slice(means, motion, quats, omega, opac, c, s, 0.5)
    assert mask is None
    o.sum().backward()  # only the opacity branch: motion gets exact zeros, scale a real gradient
    assert float(motion.grad.abs().max()) == 0.0 and float(s.grad.abs().max()) > 0 and opac.grad is not None
    with pytest.raises(RuntimeError):
        temporal_slice(means.cpu(), motion.cpu(), quats.cpu(), omega.cpu(), opac.cpu(), c.cpu(), s.cpu(), 0.5)


@pytest.mark.parametrize("ti", [0, 1, 2])
def test_temporal_slice_vs_reference_statements(ti):
    """HIP kernels against the reference's statements: within 1e-4 relative of their float64 evaluation (and no further from
    it than 4x the reference's own fp32 evaluation + 1e-6), the visibility mask equal wherever trbf is not within rounding
    of 0.05."""
    from gscodec_studio_amd.dynamic import temporal_slice

    gd = golden("dynamic.npz")
    ts = float(gd["timestamps"][ti])
    keys = ["means", "motion", "quats", "omega", "opacities", "trbf_center", "trbf_scale"]
    ins = [T(gd[k]).requires_grad_(True) for k in keys]
    m, q, o, mask = temporal_slice(*ins, ts, temp_vis_mask=True)
    ((m * T(gd["v_means_t"])).sum() + (q * T(gd["v_quats_t"])).sum() + (o * T(gd["v_opacity_t"])).sum()).backward()
    for got, name in ((m, "means_t"), (q, "quats_t"), (o, "opacity_t")):
        want, ref32 = gd[f"f64_t{ti}_{name}"], gd[f"f32_t{ti}_{name}"]
        scale = np.abs(want).max()
        err, ref_err = np.abs(N(got) - want).max(), np.abs(ref32 - want).max()
        assert err <= 1e-4 * scale and err <= 4 * ref_err + 1e-6 * scale, (name, err / scale, ref_err / scale)
    for p, k in zip(ins, keys):
        want, ref32 = gd[f"f64_t{ti}_v_{k}"], gd[f"f32_t{ti}_v_{k}"]
        scale = np.abs(want).max() + 1e-30
        err, ref_err = np.abs(N(p.grad).reshape(want.shape) - want).max(), np.abs(ref32 - want).max()
        assert err <= 1e-4 * scale and err <= 4 * ref_err + 1e-6 * scale, (k, err / scale, ref_err / scale)
    if ti == 1:
        tr = gd["f64_t1_trbf"]
        sure = np.abs(tr - 0.05) > 1e-6
        assert np.array_equal(N(mask)[sure], gd["vis_mask"][sure])
        assert np.abs(N(m)[gd["vis_mask"]] - gd["vis_means_t"]).max() <= 1e-5
