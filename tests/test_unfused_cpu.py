"""CPU: the unfused-op oracle against the golden vectors recorded from the reference's _torch_impl
(tests/golden/make_golden_unfused.py), and the indices oracle against a dense formulation."""
import numpy as np
import pytest
import torch

from util import golden

from oracle import unfused_oracle as UO


def test_world_to_cam_oracle_vs_reference():
    gd = golden("unfused.npz")
    (mc, cc), g = UO.with_grads(UO.world_to_cam, (gd["means"], gd["covars"], gd["viewmats"]), (gd["w2c.v_means_c"], gd["w2c.v_covars_c"]))
    for got, key in ((mc, "means_c"), (cc, "covars_c"), (g[0], "v_means"), (g[1], "v_covars"), (g[2], "v_viewmats")):
        ref = gd[f"w2c.{key}"]
        assert np.abs(got - ref).max() <= 2e-5 * np.abs(ref).max(), key


@pytest.mark.parametrize("model", ["pinhole", "ortho", "fisheye"])
def test_proj_oracle_vs_reference(model):
    gd = golden("unfused.npz")
    Ks = torch.tensor(gd["Ks"], dtype=torch.float64)
    W, H = int(gd["width"]), int(gd["height"])
    (m2, c2), g = UO.with_grads(lambda a, b: UO.proj(a, b, Ks, W, H, model), (gd["proj.means"], gd["proj.covars"]),
                                (gd["proj.v_means2d"], gd["proj.v_covars2d"]))
    for got, key in ((m2, "means2d"), (c2, "covars2d"), (g[0], "v_means"), (g[1], "v_covars")):
        ref = gd[f"proj.{model}.{key}"]
        bad = np.abs(got - ref) > 1e-4 * np.abs(ref) + 1e-4 * np.abs(ref).mean()
        assert bad.mean() < 2e-3, key


def test_indices_oracle_matches_dense_formulation():
    """Full-range call == every (pixel, splat) pair a front-to-back walk composites; split ranges concatenate."""
    rng = np.random.default_rng(3)
    C, N, W, H, ts = 1, 40, 24, 16, 8
    means2d = rng.uniform([0, 0], [W, H], size=(C, N, 2)).astype(np.float32)
    conics = np.tile(np.array([0.08, 0.01, 0.06], np.float32), (C, N, 1))
    opac = rng.uniform(0.3, 1.0, size=(C, N)).astype(np.float32)
    th, tw = H // ts, W // ts
    # every splat in every tile, front-to-back = index order
    flatten = np.tile(np.arange(N, dtype=np.int32), th * tw)
    offsets = (np.arange(th * tw, dtype=np.int32) * N).reshape(C, th, tw)
    T0 = np.ones((C, H, W), np.float32)
    g, p, c = UO.rasterize_to_indices_in_range(0, 10**10, T0, means2d, conics, opac, W, H, ts, offsets, flatten)
    assert len(g) > 0 and np.all(c == 0)
    # dense check of one pixel
    for pix in (0, 5 * W + 7, H * W - 1):
        i, j = divmod(pix, W)
        T = 1.0
        exp_ids = []
        for n in range(N):
            dx, dy = means2d[0, n, 0] - (j + 0.5), means2d[0, n, 1] - (i + 0.5)
            s = 0.5 * (0.08 * dx * dx + 0.06 * dy * dy) + 0.01 * dx * dy
            a = min(0.999, opac[0, n] * np.exp(-s))
            if s < 0 or a < 1 / 255:
                continue
            if T * (1 - a) <= 1e-4:
                break
            exp_ids.append(n)
            T *= 1 - a
        assert list(g[p == pix]) == exp_ids
    # one batch = ts*ts = 64 entries >= N: a single batch per tile; range [1, 2) is empty
    g2, _, _ = UO.rasterize_to_indices_in_range(1, 2, T0, means2d, conics, opac, W, H, ts, offsets, flatten)
    assert len(g2) == 0


def test_temporal_slice_oracle_finite_differences():
    """The slicing oracle (parity unpinned by reference code) against central differences of itself, incl. the
    detached tau: trbf_center must get gradient through the opacity only."""
    rs = np.random.RandomState(0)
    n = 6
    vals = [rs.randn(n, 3), 0.3 * rs.randn(n, 9), rs.randn(n, 4), 0.5 * rs.randn(n, 4), rs.uniform(0.1, 1, n),
            rs.uniform(0, 1, (n, 1)), np.exp(rs.uniform(-1, 0.5, (n, 1)))]
    vm, vq, vo = rs.randn(n, 3), rs.randn(n, 4), rs.randn(n)
    t = 0.4
    _, grads = UO.with_grads(lambda *a: UO.temporal_slice(*a, t)[:3], vals, (vm, vq, vo))

    def loss(vs, detach_check=False):
        ins = [torch.tensor(v, dtype=torch.float64) for v in vs]
        m, q, o, _ = UO.temporal_slice(*ins, t)
        return float((m * torch.tensor(vm)).sum() + (q * torch.tensor(vq)).sum() + (o * torch.tensor(vo)).sum())

    eps = 1e-6
    for which, idx in [(1, (2, 4)), (3, (1, 2)), (4, (3,)), (6, (0, 0))]:  # motion, omega, opacities, trbf_scale
        vp = [v.copy() for v in vals]
        vm_ = [v.copy() for v in vals]
        vp[which][idx] += eps
        vm_[which][idx] -= eps
        fd = (loss(vp) - loss(vm_)) / (2 * eps)
        assert abs(fd - grads[which][idx]) <= 1e-5 * (abs(fd) + 1e-3), (which, fd, grads[which][idx])
    # centre: the true derivative of the loss includes the motion/rotation paths; the (detached) oracle gradient is the
    # opacity path alone, which is what the trainer trains with
    ins = [torch.tensor(v, dtype=torch.float64) for v in vals]
    tau = t - ins[5].reshape(-1)
    d = tau / (2 ** 0.5 * ins[6].reshape(-1))
    want = torch.tensor(vo) * ins[4] * torch.exp(-d * d) * (-2 * d) * (-1 / (2 ** 0.5 * ins[6].reshape(-1)))
    assert np.allclose(grads[5].reshape(-1), want.numpy(), rtol=1e-9, atol=1e-12)


def test_temporal_slice_oracle_vs_reference_statements():
    """oracle/unfused_oracle.py:temporal_slice against tests/golden/dynamic.npz -- outputs and autograd gradients of the
    reference trainer's own statements (examples/simple_trainer_dyngs.py:506-536, executed by make_golden_dynamic.py)."""
    from util import golden

    gd = golden("dynamic.npz")
    keys = ["means", "motion", "quats", "omega", "opacities", "trbf_center", "trbf_scale"]
    vs = (gd["v_means_t"], gd["v_quats_t"], gd["v_opacity_t"])
    for ti, ts in enumerate(gd["timestamps"]):
        outs, grads = UO.with_grads(lambda *a: UO.temporal_slice(*a, float(ts))[:3], [gd[k] for k in keys], vs)
        for name, o in zip(("means_t", "quats_t", "opacity_t"), outs):
            assert np.abs(o - gd[f"f64_t{ti}_{name}"]).max() <= 1e-12 * max(1.0, np.abs(o).max()), name
            assert np.abs(o - gd[f"f32_t{ti}_{name}"]).max() <= 3e-5 * max(1.0, np.abs(o).max()), name
        for k, g in zip(keys, grads):
            want = gd[f"f64_t{ti}_v_{k}"]
            assert np.abs(g.reshape(want.shape) - want).max() <= 1e-12 * max(1.0, np.abs(want).max()), k
    tr = UO.temporal_slice(*[torch.tensor(gd[k], dtype=torch.float64) for k in keys], 0.3125)[3].numpy()
    sure = np.abs(tr - 0.05) > 1e-6
    assert np.array_equal((tr > 0.05)[sure], gd["vis_mask"][sure]) and int(gd["vis_count"]) == int(gd["vis_mask"].sum())


def test_accumulate_oracle_vs_per_ray_loop():
    """``UO.accumulate`` (vectorised restatement of _torch_impl.py:432-519 + nerfacc's two published definitions) against the definition
    itself, one python loop per ray: weight = alpha x prod(1 - alpha) over the entries in front, renders = sum weight colour."""
    rng = np.random.default_rng(7)
    C, N, W, H, ch = 2, 30, 6, 5, 5
    means2d = torch.tensor(rng.uniform([0, 0], [W, H], size=(C, N, 2)))
    conics = torch.tensor(np.tile(np.array([0.9, 0.1, 0.7]), (C, N, 1)) * rng.uniform(0.5, 1.5, size=(C, N, 1)))
    opac = torch.tensor(rng.uniform(0.2, 1.5, size=(C, N)))   # > 1: the 0.999 cap is exercised
    colors = torch.tensor(rng.normal(size=(C, N, ch)))
    g, p, c = [], [], []
    for cam in range(C):
        for pix in rng.permutation(W * H)[: W * H - 4]:      # some rays stay empty
            k = int(rng.integers(1, 7))
            g += list(rng.choice(N, size=k, replace=False))
            p += [int(pix)] * k
            c += [cam] * k
    order = np.lexsort((np.arange(len(g)), np.array(p), np.array(c)))  # rays sorted, list order kept inside a ray
    g, p, c = (np.array(x)[order] for x in (g, p, c))
    ins = [t.clone().requires_grad_(True) for t in (means2d, conics, opac, colors)]
    r, a = UO.accumulate(*ins, g, p, c, W, H)
    vr, va = torch.tensor(rng.normal(size=r.shape)), torch.tensor(rng.normal(size=a.shape))
    grads = torch.autograd.grad((r * vr).sum() + (a * va).sum(), ins)

    ins2 = [t.clone().requires_grad_(True) for t in (means2d, conics, opac, colors)]
    m2, cn, op, col = ins2
    r2 = torch.zeros((C, H, W, ch), dtype=torch.float64)
    a2 = torch.zeros((C, H, W, 1), dtype=torch.float64)
    i = 0
    while i < len(g):
        j = i
        T = torch.tensor(1.0, dtype=torch.float64)
        cam, pix = int(c[i]), int(p[i])
        rr, aa = 0.0, 0.0
        while j < len(g) and c[j] == cam and p[j] == pix:
            dx = (pix % W) + 0.5 - m2[cam, g[j], 0]
            dy = (pix // W) + 0.5 - m2[cam, g[j], 1]
            sigma = 0.5 * (cn[cam, g[j], 0] * dx * dx + cn[cam, g[j], 2] * dy * dy) + cn[cam, g[j], 1] * dx * dy
            alpha = torch.clamp_max(op[cam, g[j]] * torch.exp(-sigma), 0.999)
            rr = rr + alpha * T * col[cam, g[j]]
            aa = aa + alpha * T
            T = T * (1 - alpha)
            j += 1
        r2[cam, pix // W, pix % W] = rr
        a2[cam, pix // W, pix % W, 0] = aa
        i = j
    grads2 = torch.autograd.grad((r2 * vr).sum() + (a2 * va).sum(), ins2)
    assert torch.allclose(r, r2, rtol=1e-12, atol=1e-13) and torch.allclose(a, a2, rtol=1e-12, atol=1e-13)
    for x, y in zip(grads, grads2):
        assert torch.allclose(x, y, rtol=1e-10, atol=1e-12)
