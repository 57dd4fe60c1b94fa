"""GPU parity of the fused temporal slicing (gs_temporal_slice_fwd/bwd) against the float64 restatement of the
trainer's formulas (oracle/unfused_oracle.py:temporal_slice; parity unpinned by executable reference code, see there)
and an eager-torch fp32 formulation of the same chain."""
import numpy as np
import pytest
import torch

from util import N, T

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
