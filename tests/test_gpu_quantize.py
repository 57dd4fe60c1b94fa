"""GPU parity of the quantize/dequantize STE kernels: BIT EXACT against the golden vectors
produced by the reference's gsplat/compression_simulation/ops.py (CPU) and the oracle."""
import numpy as np
import pytest
import torch

from util import N, T, golden

pytestmark = pytest.mark.gpu

from oracle import gs_oracle as O  # noqa: E402

BOUNDS = ["scales", "quats", "opacities", "sh0", "stg_opacities", "stg_colors", "features"]


def bits_equal(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


@pytest.mark.parametrize("name", BOUNDS)
@pytest.mark.parametrize("bits", [8, 4])
def test_round_ste_bit_exact(name, bits):
    from gscodec_studio_amd.compression_simulation import STE, fake_quantize_ste

    gd = golden("quantize.npz")
    lo, hi = gd[f"{name}_bounds"]
    lo, hi = (int(lo) if float(lo).is_integer() else float(lo)), (int(hi) if float(hi).is_integer() else float(hi))
    param = torch.nn.Parameter(T(gd[f"{name}_x"]))
    out = fake_quantize_ste(param, lo, hi, bits, "round")
    assert bits_equal(N(out["output_value"]), gd[f"{name}_round{bits}_out"])
    # the parameter itself was clamped in place (reference ops.py:63)
    assert bits_equal(N(param), gd[f"{name}_round{bits}_x_after"])
    assert out["q_step"] == (hi - lo) / (2**bits - 1)
    # identity gradient everywhere, including clamped elements
    v = torch.randn_like(param)
    (g,) = torch.autograd.grad((out["output_value"] * v).sum(), param)
    assert torch.equal(g, v)


@pytest.mark.parametrize("name", BOUNDS)
@pytest.mark.parametrize("bits", [8, 4])
def test_noise_kernels_bit_exact(name, bits):
    from gscodec_studio_amd import _backend as B

    gd = golden("quantize.npz")
    lo, hi = (float(v) for v in gd[f"{name}_bounds"])
    x, noise = T(gd[f"{name}_x"]), T(gd[f"{name}_noise{bits}_noise"])
    q_step = float(gd[f"{name}_noise{bits}_q_step"])
    out = torch.empty_like(x)
    st = torch.cuda.current_stream().cuda_stream
    B.call("gs_quantize_noise_fwd", x.numel(), B.ptr(x), B.ptr(noise), O.f32(lo), O.f32(hi), O.f32(q_step), 0, B.ptr(out), st)
    assert bits_equal(N(out), gd[f"{name}_noise{bits}_out"])
    v_out = T(gd[f"{name}_noise{bits}_v_out"])
    v_x = torch.empty_like(x)
    B.call("gs_quantize_noise_bwd", x.numel(), B.ptr(x), B.ptr(v_out), O.f32(lo), O.f32(hi), 0, None, B.ptr(v_x), st)
    assert bits_equal(N(v_x), gd[f"{name}_noise{bits}_v_x"])


def test_noise_autograd_and_rng_stream():
    """fake_quantize_ste("noise") consumes the device generator exactly like the reference's
    torch.empty_like(x).uniform_(-0.5, 0.5) and masks gradients outside [lo, hi]."""
    from gscodec_studio_amd.compression_simulation import fake_quantize_ste

    x = (torch.rand(100_003, device="cuda:0") * 4 - 2).requires_grad_(True)
    torch.manual_seed(99)
    out = fake_quantize_ste(x, -1, 1, 8)  # default q_type == "noise"
    torch.manual_seed(99)
    noise = torch.empty_like(x).uniform_(-0.5, 0.5)
    q = 2 / 255
    expect = O.quant_noise_fwd(N(x), N(noise), -1, 1, q)
    assert bits_equal(N(out["output_value"]), expect)
    (g,) = torch.autograd.grad(out["output_value"].sum(), x)
    assert torch.equal(g, ((x >= -1) & (x <= 1)).float())
    with pytest.raises(UnboundLocalError):
        fake_quantize_ste(x, -1, 1, 8, "vq")


def test_compression_simulation_hooks():
    from gscodec_studio_amd.compression_simulation import CompressionSimulation, STGCompressionSimulation

    n = 5000
    g = torch.Generator(device="cuda:0").manual_seed(0)
    splats = {
        "means": torch.randn(n, 3, device="cuda:0", generator=g), "scales": torch.randn(n, 3, device="cuda:0", generator=g) * 3 - 4,
        "quats": torch.randn(n, 4, device="cuda:0", generator=g), "opacities": torch.randn(n, device="cuda:0", generator=g) * 5,
        "sh0": torch.randn(n, 1, 3, device="cuda:0", generator=g), "shN": torch.randn(n, 15, 3, device="cuda:0", generator=g),
    }
    splats = {k: torch.nn.Parameter(v) for k, v in splats.items()}
    before = {k: v.detach().clone() for k, v in splats.items()}
    sim = CompressionSimulation(entropy_model_enable=False, entropy_steps={k: -1 for k in splats})
    new, bits = sim.simulate_compression(splats, step=100)
    assert set(new) == set(splats) and all(v is None for v in bits.values())
    assert torch.equal(new["means"], before["means"]) and new["means"] is not splats["means"]
    assert new["shN"] is splats["shN"]
    for k, (lo, hi) in dict(scales=(-10, 2), quats=(-1, 1), opacities=(-15, 15), sh0=(-2, 4)).items():
        q = (hi - lo) / 255
        err = (new[k].detach() - before[k].clamp(lo, hi)).abs().max()
        assert float(err) <= q / 2 * 1.0001, k  # uniform noise of +- q/2 around the clamped value
        assert torch.equal(splats[k].detach(), before[k])  # noise mode leaves the parameter alone
    # dynamic variant, round mode: parameters ARE clamped in place
    names = ("means", "scales", "quats", "opacities", "trbf_center", "trbf_scale", "motion", "omega", "colors",
             "features_dir", "features_time")
    dims = (3, 3, 4, 1, 1, 1, 9, 4, 3, 3, 3)
    dyn = {k: torch.nn.Parameter(torch.randn(n, d, device="cuda:0", generator=g) * 6) for k, d in zip(names, dims)}
    sim2 = STGCompressionSimulation(quantization_sim_type="round", entropy_steps={})
    new2, _ = sim2.simulate_compression(dyn, step=0)
    for k, (lo, hi) in dict(scales=(-10, 2), quats=(-1, 1), opacities=(-7, 7), colors=(-7.5, 7.5),
                            features_dir=(-10, 10), features_time=(-10, 10)).items():
        assert float(dyn[k].detach().min()) >= lo and float(dyn[k].detach().max()) <= hi
        lv = (new2[k] - lo) / ((hi - lo) / 255)
        assert float((lv - lv.round()).detach().abs().max()) < 1e-3
    with pytest.raises(NotImplementedError):  # the hash-grid Gaussian model is not built (the factorized prior is: test_gpu_entropy.py)
        CompressionSimulation(entropy_model_enable=True, entropy_model_type="gaussian_model", entropy_steps={"scales": 1})



@pytest.mark.parametrize("q_type", ["noise", "round"])
@pytest.mark.parametrize("act", ["exp", "sigmoid"])
def test_fused_activation_matches_hook_then_torch_activation(q_type, act):
    """Opt-in fusion (not in the reference): fake_quantize_ste(..., activation=act) == act(fake_quantize_ste(...)) -- the same
    noise from the same generator state, the quantized value bit-identical underneath (recovered through the inverse
    activation to rounding), and the gradient the chain rule of the two-step form gives."""
    from gscodec_studio_amd.compression_simulation import fake_quantize_ste

    lo, hi = (-10, 2) if act == "exp" else (-15, 15)
    g = torch.Generator(device="cuda").manual_seed(7)
    x0 = (torch.rand(100_003, device="cuda", generator=g) * (hi - lo) * 1.2 + lo * 1.1).contiguous()  # some values outside the bounds
    v = torch.randn(100_003, device="cuda", generator=g)
    f = torch.exp if act == "exp" else torch.sigmoid

    xa = x0.clone().requires_grad_(True)
    torch.manual_seed(99)
    two = f(fake_quantize_ste(xa, lo, hi, 8, q_type)["output_value"])
    (ga,) = torch.autograd.grad((two * v).sum(), xa)
    xb = x0.clone().requires_grad_(True)
    torch.manual_seed(99)
    one = fake_quantize_ste(xb, lo, hi, 8, q_type, activation=act)["output_value"]
    (gb,) = torch.autograd.grad((one * v).sum(), xb)
    assert torch.allclose(one, two, rtol=2e-6, atol=1e-30)
    assert torch.allclose(gb, ga, rtol=1e-5, atol=1e-12)
    if q_type == "noise":  # gradient mask of the clamp: exact zeros outside the bounds
        out_of_range = (x0 < lo) | (x0 > hi)
        assert int(out_of_range.sum()) > 0 and float(gb[out_of_range].abs().max()) == 0.0
    else:  # the parameter was clamped in place in both forms
        assert torch.equal(xa.detach(), xb.detach()) and float(xb.min()) >= lo and float(xb.max()) <= hi


@pytest.mark.parametrize("sizes", [(1,), (5, 1000), (4099, 3, 777, 100_003), (3_018_195, 4_024_260, 1_006_065, 3_018_195), (2**22 + 17, 12)])
def test_multi_tensor_noise_quantizer_replays_torch_rng(sizes):
    """gs_quantize_noise_multi_fwd generates its noise IN the kernel from the default generator's (seed, offset), draw for draw
    what ``torch.empty_like(x).uniform_(-0.5, 0.5)`` would have produced tensor after tensor (Philox4x32-10, torch's grid and
    counter scheme): outputs BIT-IDENTICAL to the tensor-by-tensor hooks, the generator left at the same offset, gradients equal."""
    from gscodec_studio_amd.compression_simulation import fake_quantize_ste
    from gscodec_studio_amd.compression_simulation.ops import fake_quantize_noise_multi

    g = torch.Generator(device="cuda:0").manual_seed(3)
    bounds = [(-10, 2), (-1, 1), (-15, 15), (-2, 4)]
    acts = [None, None, "sigmoid", "exp"]
    xs0 = [torch.randn(n, device="cuda:0", generator=g) * 3 for n in sizes]
    b = [bounds[i % 4] for i in range(len(sizes))]
    a = [acts[i % 4] for i in range(len(sizes))]
    gen = torch.cuda.default_generators[0]
    torch.manual_seed(4242)
    torch.rand(7, device="cuda:0")  # (a non-zero starting offset)
    xs1 = [x.clone().requires_grad_(True) for x in xs0]
    one = [fake_quantize_ste(x, lo, hi, 8, "noise", activation=ac)["output_value"] for x, (lo, hi), ac in zip(xs1, b, a)]
    off_one = gen.get_offset()
    after_one = torch.rand(5, device="cuda:0")
    torch.manual_seed(4242)
    torch.rand(7, device="cuda:0")
    xs2 = [x.clone().requires_grad_(True) for x in xs0]
    multi = fake_quantize_noise_multi(xs2, b, [8] * len(sizes), a)
    assert gen.get_offset() == off_one
    assert torch.equal(torch.rand(5, device="cuda:0"), after_one)  # the stream continues identically
    for o, m, (lo, hi) in zip(one, multi, b):
        assert bits_equal(N(m["output_value"]), N(o)) and m["q_step"] == (hi - lo) / 255
    vs = [torch.randn_like(x) for x in xs0]
    g1 = torch.autograd.grad(sum((o * v).sum() for o, v in zip(one, vs)), xs1)
    g2 = torch.autograd.grad(sum((m["output_value"] * v).sum() for m, v in zip(multi, vs)), xs2)
    for p, q in zip(g1, g2):
        assert torch.equal(p, q)


def test_hooks_in_one_launch_equal_hooks_one_by_one():
    """CompressionSimulation.simulate_compression quantizes all hooked attributes in one launch (ops.fake_quantize_noise_multi);
    with GS_QUANT_MULTI off it runs them one by one like the reference: identical values, identical RNG stream, and a
    subset of gradients requested."""
    from gscodec_studio_amd.compression_simulation import CompressionSimulation

    n = 20_011
    g = torch.Generator(device="cuda:0").manual_seed(0)
    raw = {"means": torch.randn(n, 3, device="cuda:0", generator=g), "scales": torch.randn(n, 3, device="cuda:0", generator=g) * 3 - 4,
           "quats": torch.randn(n, 4, device="cuda:0", generator=g), "opacities": torch.randn(n, device="cuda:0", generator=g) * 5,
           "sh0": torch.randn(n, 1, 3, device="cuda:0", generator=g), "shN": torch.randn(n, 15, 3, device="cuda:0", generator=g)}
    res = {}
    for multi in (True, False):
        for activate in (False, True):
            sim = CompressionSimulation(entropy_model_enable=False, entropy_steps={k: -1 for k in raw})
            sim._MULTI = multi
            splats = {k: v.clone().requires_grad_(k != "quats") for k, v in raw.items()}
            torch.manual_seed(77)
            new, _ = sim.simulate_compression(splats, step=5, activate=activate)
            tail = torch.rand(3, device="cuda:0")
            loss = sum((new[k] * (i + 1)).sum() for i, k in enumerate(("scales", "quats", "opacities", "sh0")))
            loss.backward()
            res[(multi, activate)] = ({k: new[k].detach() for k in new}, tail, {k: p.grad for k, p in splats.items() if p.grad is not None})
    for activate in (False, True):
        (a, ta, ga), (b, tb, gb) = res[(True, activate)], res[(False, activate)]
        assert torch.equal(ta, tb)
        for k in a:
            assert torch.equal(a[k], b[k]), (k, activate)
        assert set(ga) == set(gb) == {"scales", "opacities", "sh0"}
        for k in ga:
            assert torch.equal(ga[k], gb[k]), (k, activate)


def test_round_hooks_in_one_launch_equal_hooks_one_by_one():
    """STGCompressionSimulation("round"): all hooked attributes through ONE launch (ops.fake_quantize_round_multi, round 6) against the
    tensor-by-tensor STE calls of the reference (GS_QUANT_MULTI off): identical quantized values, identical in-place clamps of the
    parameters, identical gradients -- with and without the fused activations -- and a ragged / unaligned tensor in the set."""
    from gscodec_studio_amd.compression_simulation import STGCompressionSimulation

    n = 20_011
    g = torch.Generator(device="cuda:0").manual_seed(3)
    R = lambda *s: torch.randn(*s, device="cuda:0", generator=g)  # noqa: E731
    raw = {"means": R(n, 3), "scales": R(n, 3) * 4 - 4, "quats": R(n, 4) * 0.8, "opacities": R(n) * 6, "trbf_center": R(n, 1),
           "trbf_scale": R(n, 1), "motion": R(n, 9), "omega": R(n, 4), "colors": R(n, 3) * 5, "features_dir": R(n, 3) * 7,
           "features_time": R(n, 3) * 7}
    res = {}
    for multi in (True, False):
        for activate in (False, True):
            sim = STGCompressionSimulation(quantization_sim_type="round", entropy_steps={})
            sim._MULTI = multi
            splats = {k: v.clone().requires_grad_(k != "quats") for k, v in raw.items()}
            new, _ = sim.simulate_compression(splats, step=5, activate=activate)
            loss = sum((new[k] * (i + 1)).sum() for i, k in enumerate(("scales", "quats", "opacities", "colors", "features_time", "motion")))
            loss.backward()
            res[(multi, activate)] = ({k: new[k].detach() for k in new}, {k: p.detach().clone() for k, p in splats.items()},
                                      {k: p.grad for k, p in splats.items() if p.grad is not None})
    for activate in (False, True):
        (a, pa, ga), (b, pb, gb) = res[(True, activate)], res[(False, activate)]
        for k in a:
            assert torch.equal(a[k], b[k]), (k, activate)
        for k in pa:  # the parameters after the hooks: clamped in place where hooked, untouched elsewhere
            assert torch.equal(pa[k], pb[k]), (k, activate)
        assert float(pa["scales"].max()) <= 2.0 and float(pa["features_dir"].min()) >= -10.0 and torch.equal(pa["motion"], raw["motion"])
        assert not torch.equal(pa["scales"], raw["scales"])  # (something was out of range)
        assert set(ga) == set(gb)
        for k in ga:
            assert torch.equal(ga[k], gb[k]), (k, activate)
    # a non-contiguous hooked tensor keeps the per-tensor route (STE refuses it, as before)
    sim = STGCompressionSimulation(quantization_sim_type="round", entropy_steps={})
    bad = {k: v.clone() for k, v in raw.items()}
    bad["colors"] = torch.randn(3, n, device="cuda:0").t()
    with pytest.raises(RuntimeError):
        sim.simulate_compression(bad, step=0)
