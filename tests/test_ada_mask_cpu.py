"""CPU: oracle/ada_mask_oracle.py against the fixtures generated from the reference's AnnealingMask /
CompressionSimulation (tests/golden/make_golden_ada_mask.py), and the host-side surface of the product classes
(attributes the reference trainers read, constructor errors) -- no compute calls without a GPU."""
import numpy as np
import pytest
import torch

from util import assert_close, golden

from oracle import ada_mask_oracle as AO

CASES = ["deg3", "deg1", "deg2", "deg4"]
ENTROPY_STEPS = {"means": -1, "quats": 10_000, "scales": 10_000, "opacities": 10_000, "sh0": 20_000, "shN": 10_000}


def test_temperature_schedule():
    gd = golden("ada_mask.npz")
    for s, t in zip(gd["schedule_steps"], gd["schedule_temperature"]):
        assert AO.temperature(int(s)) == float(t)
    from gscodec_studio_amd.compression_simulation import AnnealingMask

    m = AnnealingMask(input_shape=[4, 1, 1], device="cpu", annealing_start_iter=10_000)
    for s, t in zip(gd["schedule_steps"], gd["schedule_temperature"]):
        assert m.get_temperature(int(s)) == float(t)
    assert m.mask_logits.shape == (4, 1, 1) and bool((m.mask_logits == 1).all()) and m.training and m.current_iter == 0


@pytest.mark.parametrize("name", CASES)
def test_oracle_vs_reference(name):
    gd = golden("ada_mask.npz")
    x, lg, v_out = gd[f"{name}_x"], gd[f"{name}_logits"], gd[f"{name}_v_out"]
    for step in gd[f"{name}_steps"]:
        T = float(gd[f"{name}_s{step}_temperature"])
        assert T == AO.temperature(int(step))
        assert_close(AO.mask_forward(x, lg, T, True), gd[f"{name}_s{step}_train_out"], 1e-6, 1e-12, "train out")
        v_x, v_l = AO.mask_backward(x, lg, T, v_out, True)
        assert_close(v_x, gd[f"{name}_s{step}_train_v_x"], 1e-6, 1e-12, "train v_x")
        assert_close(v_l, gd[f"{name}_s{step}_train_v_logits"], 4e-6, 2e-7, "train v_logits")
        loss, v_sl = AO.sparsity_loss(lg, T)
        assert_close(loss, gd[f"{name}_s{step}_sparsity_loss"], 1e-6, 0, "sparsity loss")
        assert_close(v_sl, gd[f"{name}_s{step}_sparsity_v_logits"], 5e-6, 1e-12, "sparsity v_logits")
    assert np.array_equal(AO.mask_forward(x, lg, 1.0, False), gd[f"{name}_eval_out"])
    assert np.array_equal(AO.mask_backward(x, lg, 1.0, v_out, False)[0], gd[f"{name}_eval_v_x"])
    assert np.array_equal(AO.binary_mask(lg), gd[f"{name}_binary_mask"])
    assert AO.mask_ratio(lg) == gd[f"{name}_mask_ratio"]


def test_oracle_gradient_threshold():
    gd = golden("ada_mask.npz")
    for tag in ("mostly_zero", "few_zero"):
        got = AO.shn_gradient_threshold(gd[f"thr_{tag}_param"], gd[f"thr_{tag}_grad_in"])
        assert np.array_equal(got, gd[f"thr_{tag}_grad_out"])


def test_trainer_facing_attributes_exist():
    """What examples/simple_trainer.py:619-632, 1006, 1046-1050, 1093-1103, 1150-1162 and simple_trainer_dyngs.py:406-411,
    771 dereference on the simulation object -- with the mask off, on, and in the "gradient" strategy."""
    from gscodec_studio_amd.compression_simulation import AnnealingMask, CompressionSimulation, STGCompressionSimulation

    off = CompressionSimulation(False, "factorized_model", ENTROPY_STEPS, "cpu", False, 10_000, "learnable", cap_max=100)
    assert off.shN_ada_mask_opt is False and off.shN_qat is False and off.shN_ada_mask_step == 10_000
    assert off.shN_ada_mask_strategy == "learnable" and not hasattr(off, "shN_ada_mask")
    on = CompressionSimulation(True, "factorized_model", ENTROPY_STEPS, "cpu", True, 7_000, "learnable", cap_max=123)
    assert on.shN_ada_mask_opt is True and isinstance(on.shN_ada_mask, AnnealingMask)
    assert on.shN_ada_mask.mask_logits.shape == (123, 1, 1) and on.shN_ada_mask.annealing_start_iter == 7_000
    opt = on.shN_ada_mask_optimizer
    assert isinstance(opt, torch.optim.Adam) and opt.param_groups[0]["lr"] == 0.01
    assert opt.param_groups[0]["params"][0] is on.shN_ada_mask.mask_logits
    assert set(on.entropy_models) == set(on.entropy_model_optimizers) == set(on.entropy_model_schedulers) == set(ENTROPY_STEPS)
    assert on.entropy_min_step == 10_000
    default_cap = CompressionSimulation(False, "factorized_model", ENTROPY_STEPS, "cpu", True)
    assert default_cap.shN_ada_mask.mask_logits.shape == (1_000_000, 1, 1)  # kwargs.get("cap_max", 1_000_000)
    grad = CompressionSimulation(False, "factorized_model", ENTROPY_STEPS, "cpu", True, 10_000, "gradient", cap_max=5)
    assert grad.shN_ada_mask_opt and not hasattr(grad, "shN_ada_mask") and callable(grad.shN_gradient_threshold)
    with pytest.raises(ValueError):
        CompressionSimulation(False, "factorized_model", ENTROPY_STEPS, "cpu", True, 10_000, None)
    with pytest.raises(NotImplementedError):
        CompressionSimulation(False, "factorized_model", ENTROPY_STEPS, "cpu", True, 10_000, "topk")
    stg = STGCompressionSimulation("round", False, None, "cpu", True, 9_000, cap_max=77)
    assert stg.shN_ada_mask_opt and stg.shN_qat is False and stg.shN_ada_mask_step == 9_000
    assert stg.shN_ada_mask.mask_logits.shape == (77, 1, 1) and isinstance(stg.shN_ada_mask_optimizer, torch.optim.Adam)
    stg_off = STGCompressionSimulation("round", False, None, "cpu")
    assert stg_off.shN_ada_mask_opt is False and not hasattr(stg_off, "shN_ada_mask")


def test_cpu_tensors_raise():
    from gscodec_studio_amd.compression_simulation import AnnealingMask

    m = AnnealingMask(input_shape=[8, 1, 1], device="cpu")
    with pytest.raises(RuntimeError):
        m(torch.zeros(8, 15, 3), 20_000)
    with pytest.raises(RuntimeError):
        m.get_sparsity_loss()
