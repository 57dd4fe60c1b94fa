"""GPU parity of the shN adaptive mask (csrc/ada_mask.hip behind compression_simulation/ada_mask.py) against the fixtures
generated from the reference's AnnealingMask / CompressionSimulation (tests/golden/make_golden_ada_mask.py), the
"trainer contract" of the compression-simulation object (every attribute read and call the reference's trainer makes on
it, examples/simple_trainer.py:619-632, 906-907, 991-1007, 1046-1050, 1070-1074, 1093-1103, 1150-1162, restated as an
access pattern), and size-independent properties at 1 M splats.

Tolerances: the mask value goes through one exp and two IEEE operations -> 1e-6 relative; given the mask the products are
exact; the per-splat logit gradient is a 3 (K - 1)-term sum -> 4e-6 relative + 2e-7 absolute (the bounds the oracle is held to
against the reference).  The eval-mode (binary) outputs, the binary mask, the mask ratio and the gradient threshold are
compared bit-exact."""
import numpy as np
import pytest
import torch

from util import N, T, assert_close, dev, garden, garden_sh, golden

pytestmark = pytest.mark.gpu

from oracle import ada_mask_oracle as AO  # noqa: E402

CASES = ["deg3", "deg1", "deg2", "deg4"]
ENTROPY_STEPS = {"means": -1, "quats": 10_000, "scales": 10_000, "opacities": 10_000, "sh0": 20_000, "shN": 10_000}


def _mask(gd, name):
    from gscodec_studio_amd.compression_simulation import AnnealingMask

    lg = gd[f"{name}_logits"]
    m = AnnealingMask(input_shape=[len(lg), 1, 1], device=dev(), annealing_start_iter=10_000)
    with torch.no_grad():
        m.mask_logits.copy_(T(lg).reshape(-1, 1, 1))
    return m


@pytest.mark.parametrize("name", CASES)
def test_mask_vs_reference(name):
    gd = golden("ada_mask.npz")
    m = _mask(gd, name)
    x_np, v_out = gd[f"{name}_x"], T(gd[f"{name}_v_out"])
    for step in gd[f"{name}_steps"]:
        step = int(step)
        m.train()
        x = T(x_np, requires_grad=True)
        m.mask_logits.grad = None
        y = m(x, step)
        assert m.current_iter == step
        y.backward(v_out)
        assert_close(N(y), gd[f"{name}_s{step}_train_out"], 1e-6, 1e-12, f"{name} {step} train out")
        assert_close(N(x.grad), gd[f"{name}_s{step}_train_v_x"], 1e-6, 1e-12, f"{name} {step} train v_x")
        assert m.mask_logits.grad.shape == m.mask_logits.shape
        assert_close(N(m.mask_logits.grad).reshape(-1), gd[f"{name}_s{step}_train_v_logits"], 4e-6, 2e-7, f"{name} {step} v_logits")
        # sparsity loss + gradient (ada_mask.py:46-58)
        m.mask_logits.grad = None
        loss = m.get_sparsity_loss()
        loss.backward()
        assert loss.shape == ()
        assert_close(N(loss), gd[f"{name}_s{step}_sparsity_loss"], 2e-6, 0, f"{name} {step} sparsity loss")
        assert_close(N(m.mask_logits.grad).reshape(-1), gd[f"{name}_s{step}_sparsity_v_logits"], 5e-6, 1e-12, f"{name} {step} sparsity grad")
    # eval mode: binary mask, no gradient to the logits -- bit exact
    m.eval()
    x = T(x_np, requires_grad=True)
    m.mask_logits.grad = None
    y = m(x, 12_345)
    y.backward(v_out)
    assert m.mask_logits.grad is None
    assert np.array_equal(N(y), gd[f"{name}_eval_out"])
    assert np.array_equal(N(x.grad), gd[f"{name}_eval_v_x"])
    b = m.get_binary_mask()
    assert b.shape == m.mask_logits.shape and np.array_equal(N(b).reshape(-1), gd[f"{name}_binary_mask"])
    r = m.get_mask_ratio()
    assert r.shape == () and N(r) == gd[f"{name}_mask_ratio"]


def test_mask_is_deterministic_and_handles_layouts():
    gd = golden("ada_mask.npz")
    m = _mask(gd, "deg3")
    v_out = T(gd["deg3_v_out"])
    grads = []
    for _ in range(2):
        x = T(gd["deg3_x"], requires_grad=True)
        m.mask_logits.grad = None
        m(x, 15_000).backward(v_out)
        grads.append((N(x.grad), N(m.mask_logits.grad)))
    assert np.array_equal(grads[0][0], grads[1][0]) and np.array_equal(grads[0][1], grads[1][1])
    # non-contiguous x (a slice of a wider tensor), x without grad, logits without grad
    wide = T(np.concatenate([np.zeros_like(gd["deg3_x"][:, :1]), gd["deg3_x"]], 1))
    y = m(wide[:, 1:], 15_000)
    assert_close(N(y), gd["deg3_s15000_train_out"], 1e-6, 1e-12, "strided input")
    m.mask_logits.requires_grad_(False)
    x = T(gd["deg3_x"], requires_grad=True)
    m(x, 15_000).backward(v_out)
    assert np.array_equal(N(x.grad), grads[0][0])
    m.mask_logits.requires_grad_(True)
    # the broadcast rule of the reference: N must equal cap_max
    with pytest.raises(RuntimeError, match="must match the size"):
        m(torch.zeros(10, 15, 3, device=dev()), 15_000)
    # zero splats
    from gscodec_studio_amd.compression_simulation import AnnealingMask

    e = AnnealingMask(input_shape=[0, 1, 1], device=dev())
    assert e(torch.zeros(0, 15, 3, device=dev()), 20_000).shape == (0, 15, 3)


def test_hook_vs_reference():
    """CompressionSimulation.simulate_compression_shN (simulation.py:319-324): identity up to ada_mask_step, masked after."""
    from gscodec_studio_amd.compression_simulation import CompressionSimulation

    gd = golden("ada_mask.npz")
    shN, lg = gd["sim_shN"], gd["sim_logits"]
    sim = CompressionSimulation(False, "factorized_model", ENTROPY_STEPS, dev(), True, 10_000, "learnable", cap_max=len(lg))
    with torch.no_grad():
        sim.shN_ada_mask.mask_logits.copy_(T(lg).reshape(-1, 1, 1))
    p = T(shN)
    for step in (10_000, 10_001, 22_222):
        y, bits = sim.simulate_compression_shN(p, step, None, None)
        assert bits is None
        if step <= 10_000:
            assert y is p
        assert_close(N(y), gd[f"sim_s{step}_out"], 1e-6, 1e-12, f"hook step {step}")
    # the trainer's sequence on the mask (simple_trainer.py:1006-1007, 1160-1162): render-term gradient + sparsity loss ->
    # backward -> Adam step -> zero_grad; pinned against the reference's optimizer result
    step = 15_000
    x = T(shN, requires_grad=True)
    y, _ = sim.simulate_compression_shN(x, step, None, None)
    loss = (y * T(gd["adam_v_out"])).sum() + sim.shN_ada_mask.get_sparsity_loss()
    loss.backward()
    assert_close(N(sim.shN_ada_mask.mask_logits.grad).reshape(-1), gd["adam_grad"], 5e-6, 2e-9, "mask gradient")
    sim.shN_ada_mask_optimizer.step()
    sim.shN_ada_mask_optimizer.zero_grad(set_to_none=True)
    assert sim.shN_ada_mask.mask_logits.grad is None
    # Adam's first step moves every logit by lr * sign(g) (up to eps): compare to the reference's result
    assert_close(N(sim.shN_ada_mask.mask_logits).reshape(-1), gd["adam_logits_after"], 1e-6, 1e-6, "logits after Adam")


def test_gradient_threshold_bit_exact():
    from gscodec_studio_amd.compression_simulation import CompressionSimulation

    gd = golden("ada_mask.npz")
    sim = CompressionSimulation(False, "factorized_model", ENTROPY_STEPS, dev(), True, 10_000, "gradient", cap_max=5)
    for tag in ("mostly_zero", "few_zero"):
        par = torch.nn.Parameter(T(gd[f"thr_{tag}_param"]))
        par.grad = T(gd[f"thr_{tag}_grad_in"])
        sim.shN_gradient_threshold(par, 12_345)
        assert np.array_equal(N(par.grad), gd[f"thr_{tag}_grad_out"]), tag
        assert np.array_equal(N(par), gd[f"thr_{tag}_param"])


def test_trainer_contract():
    """The reference trainer's access pattern on the simulation object for the flag set of
    examples/benchmarks/compression/mcmc_tt_sim.sh:31-33 (--compression_sim --entropy_model_opt --shN_ada_mask_opt):
    construct with the positional arguments of simple_trainer.py:619-626, then per step the reads / calls of lines 906-907,
    779-786 (activations), 991-1007, 1046-1050, 1070-1074, 1093-1095, 1150-1162 -- around a real render."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd.compression_simulation import CompressionSimulation

    g = garden(1500, scale_mult=4.0)
    n = len(g["means"])
    sh = garden_sh(g["rgb"])
    entropy_steps = {"means": -1, "quats": 1, "scales": 1, "opacities": 1, "sh0": 3, "shN": 1}
    ada_mask_steps, strategy, rd_lambda = 2, "learnable", 1e-2
    sim = CompressionSimulation(True, "factorized_model", entropy_steps, dev(), True, ada_mask_steps, strategy, cap_max=n)
    entropy_min_step = entropy_steps[min((k for k, v in entropy_steps.items() if v > 0), key=lambda k: entropy_steps[k])]
    splats = torch.nn.ParameterDict({
        "means": torch.nn.Parameter(T(g["means"])), "scales": torch.nn.Parameter(T(np.log(g["scales"]))),
        "quats": torch.nn.Parameter(T(g["quats"])), "opacities": torch.nn.Parameter(torch.logit(T(g["opacities"]).clamp(0.01, 0.99))),
        "sh0": torch.nn.Parameter(T(sh[:, :1])), "shN": torch.nn.Parameter(T(sh[:, 1:]))})
    opt = torch.optim.Adam(splats.parameters(), lr=1e-3)
    viewmats, Ks = T(g["viewmats"][:1]), T(g["Ks"][:1])
    target = torch.rand(1, g["height"], g["width"], 3, device=dev())
    ratios = []
    for step in range(6):
        comp, esti_bits = sim.simulate_compression(splats, step)
        assert set(comp) == set(splats.keys()) == set(esti_bits)
        colors = torch.cat([comp["sh0"], comp["shN"]], 1)
        renders, alphas, info = rasterization(comp["means"], comp["quats"], torch.exp(comp["scales"]), torch.sigmoid(comp["opacities"]),
                                              colors, viewmats, Ks, g["width"], g["height"], sh_degree=3, packed=False)
        loss = torch.nn.functional.l1_loss(renders, target)
        if step > entropy_min_step:
            total = 0
            for k, k_step in entropy_steps.items():
                if step > k_step and esti_bits[k] is not None:
                    total = total + torch.sum(esti_bits[k]) / esti_bits[k].numel()
            loss = loss + rd_lambda * total
        masked = sim.shN_ada_mask_opt and strategy == "learnable" and step > ada_mask_steps
        if masked:
            loss = loss + sim.shN_ada_mask.get_sparsity_loss()
            assert not torch.equal(comp["shN"], splats["shN"])
        else:
            assert comp["shN"] is splats["shN"]  # the reference returns the parameter itself (simulation.py:319-324)
        loss.backward()
        assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in splats.values())
        if masked:
            ratios.append(float(sim.shN_ada_mask.get_mask_ratio()))
            assert sim.shN_ada_mask.mask_logits.grad is not None and float(sim.shN_ada_mask.mask_logits.grad.abs().sum()) > 0
            b = sim.shN_ada_mask.get_binary_mask()
            assert (splats["shN"].data * b).shape == splats["shN"].shape  # checkpoint-time masking (1070-1072)
        for name, m in sim.entropy_models.items():
            if m is not None:
                assert isinstance(m.state_dict(), dict)
        opt.step()
        opt.zero_grad(set_to_none=True)
        for name, o in sim.entropy_model_optimizers.items():
            if o is not None:
                o.step()
                o.zero_grad(set_to_none=True)
        for name, s in sim.entropy_model_schedulers.items():
            if s is not None and step > entropy_steps[name]:
                s.step()
        if masked:
            before = sim.shN_ada_mask.mask_logits.detach().clone()
            sim.shN_ada_mask_optimizer.step()
            sim.shN_ada_mask_optimizer.zero_grad(set_to_none=True)
            assert not torch.equal(before, sim.shN_ada_mask.mask_logits.detach())
    assert len(ratios) == 3 and all(0.0 <= r <= 1.0 for r in ratios)


def test_full_size_properties():
    """1 M splats, degree 3 (BASELINE config 3 / 4 size): linearity in x, the binary mask as a projection, the logit gradient
    against a float64 evaluation of the same sums on a sample, and the gradient-threshold kernel's bookkeeping."""
    from gscodec_studio_amd.compression_simulation import AnnealingMask
    from gscodec_studio_amd.compression_simulation.ada_mask import shN_gradient_threshold

    n = 1_000_003
    gen = torch.Generator(device=dev()).manual_seed(7)
    x = torch.randn(n, 15, 3, device=dev(), generator=gen) * 0.1
    m = AnnealingMask(input_shape=[n, 1, 1], device=dev())
    with torch.no_grad():
        m.mask_logits.copy_(torch.randn(n, 1, 1, device=dev(), generator=gen) * 3)
    step = 20_000
    y = m(x, step)
    assert torch.equal(m(2 * x, step), 2 * y)  # exact: scaling by a power of two commutes with the rounding
    T_ = m.get_temperature(step)
    # float64 evaluation of the same expression; the fp32 rounding of logit / T is amplified by |logit / T| (up to ~20 here)
    mask64 = torch.sigmoid(m.mask_logits.detach().double() / T_)
    ref = x.double() * mask64
    assert float(((y.detach().double() - ref).abs() / ref.abs().clamp_min(1e-30)).max()) < 4e-6
    m.eval()
    yb = m(x, step)
    b = m.get_binary_mask()
    assert torch.equal(yb, x * b) and torch.equal(m(yb, step), yb)
    assert abs(float(m.get_mask_ratio()) - float(b.sum()) / n) < 1e-7
    m.train()
    xg = x.clone().requires_grad_(True)
    v = torch.randn_like(x)
    m(xg, step).backward(v)
    sel = torch.arange(0, n, 997, device=dev())
    mk = torch.sigmoid(m.mask_logits.detach().reshape(-1)[sel].double() / T_)
    want = (v[sel].double() * x[sel].double()).sum((1, 2)) * mk * (1 - mk) / T_
    got = m.mask_logits.grad.reshape(-1)[sel].double()
    assert float((got - want).abs().max()) < 2e-6 * float(want.abs().max()) + 1e-7
    ref = v.double() * mask64
    assert float(((xg.grad.double() - ref).abs() / ref.abs().clamp_min(1e-30)).max()) < 4e-6
    # gradient threshold: 95 % zero rows -> threshold 2e-3
    p = x.clone()
    zero = torch.rand(n, device=dev(), generator=gen) < 0.95
    p[zero] = 0
    g = torch.randn_like(x) * (10.0 ** (torch.rand(n, 1, 1, device=dev(), generator=gen) * 6 - 6))
    g0 = g.clone()
    shN_gradient_threshold(p, g)
    norm = g0.double().pow(2).sum((1, 2)).sqrt()
    kill = zero & (norm < 2e-3)
    edge = (norm - 2e-3).abs() < 1e-8
    changed = (g != g0).any(-1).any(-1)
    assert torch.equal(changed | edge, kill | edge)
    assert bool((g[kill & ~edge] == 0).all()) and torch.equal(g[~kill & ~edge], g0[~kill & ~edge])


@pytest.mark.parametrize("step_driver", [True, False])
@pytest.mark.parametrize("training", [True, False])
def test_mask_fused_into_the_renderer_equals_masking_first(step_driver, training):
    """``rasterization(colors=(sh0, MaskedShN))`` -- the mask applied by the projection pass while it loads the coefficients,
    both gradients from its backward -- against masking first (``AnnealingMask.forward``) and rendering the masked tensor: the
    same roundings in the same order, so the image is identical and the gradients agree to the order of the compositing
    backward's float atomics."""
    from gscodec_studio_amd import _step, rasterization
    from gscodec_studio_amd.compression_simulation import AnnealingMask

    g = garden(2500, scale_mult=4.0)
    n = len(g["means"])
    sh = garden_sh(g["rgb"])
    gen = torch.Generator(device=dev()).manual_seed(5)
    m = AnnealingMask(input_shape=[n, 1, 1], device=dev(), annealing_start_iter=10)
    with torch.no_grad():
        m.mask_logits.copy_(torch.randn(n, 1, 1, device=dev(), generator=gen) * 2)
    m.train(training)
    vm, Ks = T(g["viewmats"][:2]), T(g["Ks"][:2])
    target = torch.rand(2, g["height"], g["width"], 3, device=dev(), generator=gen)
    res = {}
    prev = _step.ENABLED
    _step.ENABLED = step_driver
    try:
        for fused in (True, False):
            P = {k: T(v).requires_grad_(True) for k, v in dict(means=g["means"], quats=g["quats"], scales=g["scales"], opacities=g["opacities"],
                                                              sh0=sh[:, :1], shN=sh[:, 1:]).items()}
            m.mask_logits.grad = None
            shN = m.fused(P["shN"], 25) if fused else m(P["shN"], 25)
            rc, ra, meta = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], (P["sh0"], shN), vm, Ks, g["width"], g["height"],
                                         sh_degree=3, packed=False)
            ((rc - target) ** 2).sum().backward()
            res[fused] = (rc.detach(), {k: p.grad for k, p in P.items()}, None if m.mask_logits.grad is None else m.mask_logits.grad.clone())
    finally:
        _step.ENABLED = prev
    (rc_f, g_f, l_f), (rc_u, g_u, l_u) = res[True], res[False]
    assert torch.equal(rc_f, rc_u)
    for k in g_f:
        den = float(g_u[k].norm()) + 1e-30
        assert float((g_f[k] - g_u[k]).norm()) <= 3e-4 * den, (k, float((g_f[k] - g_u[k]).norm()) / den)
    if training:
        assert l_f is not None and l_u is not None and l_f.shape == m.mask_logits.shape
        assert float((l_f - l_u).norm()) <= 3e-4 * float(l_u.norm()), float((l_f - l_u).norm()) / float(l_u.norm())
        assert float(l_u.abs().max()) > 0
    else:
        assert l_f is None and l_u is None


def test_simulation_activate_hands_the_mask_to_the_renderer():
    """simulate_compression(activate=True) past ada_mask_step: new_splats["shN"] is a MaskedShN that rasterization() takes in
    the (sh0, shN) pair; degree-2 coefficients (27 floats per splat: not vectorisable) fall back to materialising the mask."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd.compression_simulation import CompressionSimulation
    from gscodec_studio_amd.compression_simulation.ada_mask import MaskedShN

    g = garden(1200, scale_mult=4.0)
    n = len(g["means"])
    for K, deg in ((16, 3), (9, 2)):
        sh = garden_sh(g["rgb"], K=K)
        sim = CompressionSimulation(False, "factorized_model", ENTROPY_STEPS, dev(), True, 2, "learnable", cap_max=n)
        splats = {"means": T(g["means"]), "scales": T(np.log(g["scales"])), "quats": T(g["quats"]),
                  "opacities": torch.logit(T(g["opacities"]).clamp(0.01, 0.99)), "sh0": T(sh[:, :1]), "shN": T(sh[:, 1:])}
        splats = {k: v.requires_grad_(True) for k, v in splats.items()}
        new, _ = sim.simulate_compression(splats, step=7, activate=True)
        assert isinstance(new["shN"], MaskedShN) and new["shN"].shN is splats["shN"]
        rc, _, _ = rasterization(new["means"], new["quats"], new["scales"], new["opacities"], (new["sh0"], new["shN"]), T(g["viewmats"][:1]),
                                 T(g["Ks"][:1]), g["width"], g["height"], sh_degree=deg, packed=False)
        (rc.sum() + sim.shN_ada_mask.get_sparsity_loss()).backward()
        assert splats["shN"].grad is not None and sim.shN_ada_mask.mask_logits.grad is not None
        assert float(sim.shN_ada_mask.mask_logits.grad.abs().sum()) > 0 and bool(torch.isfinite(splats["shN"].grad).all())
