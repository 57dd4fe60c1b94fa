"""Per-attribute compression-simulation hooks (quantizer part).

Mirrors ``CompressionSimulation`` (reference simulation.py:14-348) and
``STGCompressionSimulation`` (508-780): which attributes are fake-quantized, their bounds
and bit widths, and the ``simulate_compression(splats, step) -> (new_splats, esti_bits)``
contract.  ``entropy_model_enable=True`` with the factorized prior attaches the fused bits
estimator (entropy_model.py, reference wiring simulation.py:87-149, 247-316, 576-608); the
hash-grid Gaussian model needs the reference's CUDA-only ``_gridencoder`` and raises
NotImplementedError.  ``ada_mask_opt=True`` attaches the learnable shN mask (ada_mask.py; reference
simulation.py:38-41, 151-168, 319-324, 611-620) or, with ``ada_mask_strategy="gradient"``, the
gradient threshold (simulation.py:327-348).  Every attribute the reference's trainers read
(``shN_ada_mask_opt``, ``shN_ada_mask_step``, ``shN_ada_mask_strategy``, ``shN_qat``, ``shN_ada_mask``,
``shN_ada_mask_optimizer``, ``entropy_models``, ``entropy_model_optimizers``, ``entropy_model_schedulers``;
examples/simple_trainer.py:619-632, 906-907, 991-1007, 1046-1050, 1070-1074, 1093-1103, 1150-1162) exists.
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor
from typing_extensions import Literal

from .ada_mask import AnnealingMask
from .ada_mask import shN_gradient_threshold as _shN_gradient_threshold
from .entropy_model import Entropy_factorized_optimized_refactor
from .ops import QUANT_MULTI_MAX, fake_quantize_noise_multi, fake_quantize_round_multi, fake_quantize_ste, multi_selfcheck


class _SimulationBase:
    simulation_option: Dict[str, bool]
    q_bitwidth: Dict[str, Optional[int]]
    bds: Dict[str, Optional[list]]
    q_type: Optional[str]

    entropy_model_enable: bool = False
    entropy_model_option: Dict[str, bool] = {}
    entropy_models: Dict[str, Optional[torch.nn.Module]] = {}

    def _setup_entropy(self, enable: bool, model_type: str, specs: Dict[str, Optional[dict]]) -> None:
        """Factorized-prior models + their Adam optimizers (reference simulation.py:87-149, 576-608).
        ``specs``: attribute -> constructor kwargs (None: no model)."""
        self.entropy_model_enable = bool(enable)
        if not enable:
            return
        if model_type != "factorized_model":
            raise NotImplementedError(
                f"entropy_model_type={model_type!r}: only the factorized prior is built (the hash-grid Gaussian "
                "model depends on the reference's CUDA-only _gridencoder extension)")
        if self.entropy_steps is None:
            raise ValueError("entropy_model_enable=True needs entropy_steps")
        for name in list(self.entropy_model_option):  # turn off if entropy step < 0 (simulation.py:70-73)
            if self.entropy_steps.get(name, -1) < 0:
                self.entropy_model_option[name] = False
        self.entropy_models = {k: (Entropy_factorized_optimized_refactor(**kw).to(self.device) if kw is not None else None)
                               for k, kw in specs.items()}
        positive = [k for k, v in self.entropy_steps.items() if v > 0]
        self.entropy_min_step = self.entropy_steps[min(positive, key=lambda k: self.entropy_steps[k])] if positive else 0
        self.entropy_model_optimizers, self.entropy_model_schedulers = {}, {}
        for k, m in self.entropy_models.items():
            self.entropy_model_optimizers[k] = None if m is None else torch.optim.Adam(
                [{"params": p, "lr": 1e-4, "name": n} for n, p in m.named_parameters()])
            self.entropy_model_schedulers[k] = None

    def _setup_ada_mask(self, ada_mask_opt: bool, ada_mask_step: int, strategy: Optional[str], kwargs: dict) -> None:
        """reference simulation.py:38-41, 151-168 (static) / 548-550, 611-620 (dynamic; always "learnable" there)."""
        self.shN_qat = False  # the K-means QAT branch of the reference is dead code behind this constant (simulation.py:38, 76)
        self.shN_ada_mask_opt = ada_mask_opt
        self.shN_ada_mask_step = ada_mask_step
        self.shN_ada_mask_strategy = strategy
        if not ada_mask_opt:
            return
        if strategy == "learnable":
            cap_max = kwargs.get("cap_max", 1_000_000)
            self.shN_ada_mask = AnnealingMask(input_shape=[cap_max, 1, 1], device=self.device, annealing_start_iter=ada_mask_step)
            self.shN_ada_mask_optimizer = torch.optim.Adam([{"params": self.shN_ada_mask.parameters(), "lr": 0.01}])
        elif strategy == "gradient":
            pass
        elif strategy is None:
            raise ValueError("'shN_ada_mask_strategy' should not be None")
        else:
            raise NotImplementedError(f"'shN_ada_mask_strategy': {strategy} has not been implemented.")

    # the activations the trainers apply to the hooked values (reference examples/simple_trainer.py:779-786,
    # simple_trainer_dyngs.py:506-521): opt-in fusion into the quantizer pass, see simulate_compression(activate=True)
    ACTIVATIONS = {"scales": "exp", "opacities": "sigmoid"}
    _activate = False

    def _quantize(self, name: str, param: Tensor, step: int = 0, as_channels=None) -> Tuple[Tensor, Optional[Tensor]]:
        """fake-quantize ``param``; past ``entropy_steps[name]`` also estimate its bits.
        ``as_channels``: reshape of the quantized value to the model's [N, C] input (and back)."""
        lo, hi = self.bds[name]
        # both sides of the reference's `step < 10_000` branch select 8 bits
        # (simulation.py:242-245): the schedule is a no-op and q_bitwidth[name] is used.
        bits = self.q_bitwidth[name]
        estimate = (self.entropy_model_enable and self.entropy_model_option.get(name, False)
                    and step > self.entropy_steps[name] and self.entropy_models.get(name) is not None)
        act = self.ACTIVATIONS.get(name) if self._activate else None
        fused = act if not estimate else None  # (the bits estimator needs the quantized value itself: activate afterwards)
        pre = self._pre.pop(name, None)
        if pre is not None and pre[0] is param and pre[1] == fused:
            out = pre[2]  # quantized together with the step's other hooked attributes (one launch, see _simulate)
        elif self.q_type is None:
            out = fake_quantize_ste(param, lo, hi, bits, activation=fused)  # default q_type="noise"
        else:
            out = fake_quantize_ste(param, lo, hi, bits, self.q_type, activation=fused)
        value = out["output_value"]
        if estimate:
            x = value if as_channels is None else as_channels(value)
            bits_est = self.entropy_models[name](x, out["q_step"])
            if act is not None:
                value = torch.exp(value) if act == "exp" else torch.sigmoid(value)
            return value, bits_est
        return value, None

    def simulate_compression(self, splats: Dict[str, Tensor], step: int, activate: bool = False):
        """Returns (new_splats, esti_bits_dict); un-simulated attributes come back as ``p + 0.``

        ``activate=True`` (opt-in, not in the reference): ``new_splats["scales"]`` / ``["opacities"]`` come back ALREADY
        activated (exp / sigmoid, what the trainer applies next), evaluated inside the quantizer kernels -- two elementwise
        passes and their backward passes less per step -- and, when the learnable shN mask applies, ``new_splats["shN"]`` is a
        ``MaskedShN`` (the parameter + the mask) for ``rasterization(colors=(sh0, shN))`` to apply on the fly; together this is
        the fused form of the reference's ``simple_trainer.py:779-800``."""
        self._activate = bool(activate)
        try:
            return self._simulate(splats, step)
        finally:
            self._activate = False

    _pre: Dict[str, tuple] = {}
    _MULTI = os.environ.get("GS_QUANT_MULTI", "1") != "0"

    def _prequantize(self, splats: Dict[str, Tensor], step: int) -> None:
        """All quantized attributes of the step in ONE launch.  Noise mode (ops.fake_quantize_noise_multi): the noise is drawn in the
        kernel exactly as the per-attribute ``uniform_`` calls of the reference would have drawn it, in the same order, so the
        values and the RNG stream are those of the tensor-by-tensor hooks.  Round mode (ops.fake_quantize_round_multi, round 6): the
        same arithmetic and the same in-place clamp of every parameter as the per-tensor ``STE`` calls.  Only the class's own hook functions take part (a
        subclass that overrides one keeps its behaviour)."""
        self._pre = {}
        if not self._MULTI or self.q_type not in (None, "noise", "round"):
            return
        rnd = self.q_type == "round"
        names = []
        for name, p in splats.items():
            if not self.simulation_option.get(name, False) or self.bds.get(name) is None:
                continue
            fn = getattr(type(self), f"simulate_compression_{name}", None)
            if fn is None or getattr(fn, "__qualname__", "").split(".")[0] not in ("CompressionSimulation", "STGCompressionSimulation"):
                return
            if not (isinstance(p, Tensor) and p.is_cuda and p.dtype == torch.float32):
                return
            if rnd and not p.is_contiguous():  # (the round hook clamps its input in place: only the tensor itself will do)
                return
            names.append(name)
        if not 2 <= len(names) <= QUANT_MULTI_MAX or len({splats[n].device for n in names}) != 1:
            return
        if not rnd and not multi_selfcheck(splats[names[0]].device):  # (first use per device; a torch / ROCm change of the RNG kernel shows here)
            type(self)._MULTI = False
            return
        acts = []
        for n in names:
            estimate = (self.entropy_model_enable and self.entropy_model_option.get(n, False)
                        and step > self.entropy_steps[n] and self.entropy_models.get(n) is not None)
            acts.append(self.ACTIVATIONS.get(n) if (self._activate and not estimate) else None)
        outs = (fake_quantize_round_multi if rnd else fake_quantize_noise_multi)(
            [splats[n] for n in names], [tuple(self.bds[n]) for n in names], [self.q_bitwidth[n] for n in names], acts)
        self._pre = {n: (splats[n], a, o) for n, a, o in zip(names, acts, outs)}

    def _simulate(self, splats: Dict[str, Tensor], step: int):
        new_splats, esti_bits = {}, {}
        try:
            self._prequantize(splats, step)
            return self._run_hooks(splats, step, new_splats, esti_bits)
        finally:
            self._pre = {}  # (nothing pre-quantized outlives its call: an exception in a hook must not leave stale values behind)

    def _run_hooks(self, splats, step, new_splats, esti_bits):
        for name in splats.keys():
            if self.simulation_option[name]:
                fn = getattr(self, f"simulate_compression_{name}", None)
                if fn is None:
                    raise NotImplementedError(f"no simulate function for {name}")
                new_splats[name], esti_bits[name] = fn(splats[name], step)[:2]
            else:
                new_splats[name] = splats[name] + 0.0
                act = self.ACTIVATIONS.get(name) if self._activate else None
                if act is not None:  # activate=True promises activated scales / opacities whether or not they are simulated
                    new_splats[name] = torch.exp(new_splats[name]) if act == "exp" else torch.sigmoid(new_splats[name])
                esti_bits[name] = None
        return new_splats, esti_bits


class CompressionSimulation(_SimulationBase):
    """Static-scene hooks: scales / quats / opacities / sh0 fake-quantized at 8 bits with
    uniform noise (the static trainer never forwards a q_type, so "noise" it is --
    SURVEY.md quirk 12); shN passes through; means untouched."""

    def __init__(self, entropy_model_enable: bool = False,
                 entropy_model_type: Literal["factorized_model", "gaussian_model"] = "factorized_model",
                 entropy_steps: Optional[Dict[str, int]] = None, device=None, ada_mask_opt: bool = False,
                 ada_mask_step: int = 10_000, ada_mask_strategy: Optional[str] = "learnable", **kwargs) -> None:
        self.entropy_model_type = entropy_model_type
        self.entropy_steps = entropy_steps
        self.device = device
        self.q_type = None
        self.entropy_model_option = {"means": False, "scales": True, "quats": True, "opacities": False, "sh0": True,
                                     "shN": False}
        self._setup_entropy(entropy_model_enable, entropy_model_type, {
            "means": None, "scales": dict(channel=3, filters=(3, 3)), "quats": dict(channel=4), "opacities": None,
            "sh0": dict(channel=3, filters=(3, 3)), "shN": None})
        self.simulation_option = {"means": False, "scales": True, "quats": True, "opacities": True, "sh0": True,
                                  "shN": True}
        self.q_bitwidth = {"means": None, "scales": 8, "quats": 8, "opacities": 8, "sh0": 8, "shN": None}
        self.bds = {"means": None, "scales": [-10, 2], "quats": [-1, 1], "opacities": [-15, 15], "sh0": [-2, 4],
                    "shN": None}
        self._setup_ada_mask(ada_mask_opt, ada_mask_step, ada_mask_strategy, kwargs)

    def simulate_compression_scales(self, param, step, *_): return self._quantize("scales", param, step)
    def simulate_compression_quats(self, param, step, *_): return self._quantize("quats", param, step)

    def simulate_compression_opacities(self, param, step, *_):  # [N] -> [N, 1] for the model (simulation.py:289-296)
        return self._quantize("opacities", param, step, lambda v: v.unsqueeze(1))

    def simulate_compression_sh0(self, param, step, *_):  # [N, 1, 3] -> [N, 3] (simulation.py:308-314)
        return self._quantize("sh0", param, step, lambda v: v.squeeze(1))

    def simulate_compression_shN(self, param, step, *_):
        """reference simulation.py:319-324: past ``ada_mask_step`` the learnable mask multiplies the higher bands."""
        if self.shN_ada_mask_opt and self.shN_ada_mask_strategy == "learnable" and step > self.shN_ada_mask_step:
            # (activate=True, opt-in: the mask rides to the renderer, which applies it while it loads the coefficients)
            param = self.shN_ada_mask.fused(param, step) if self._activate else self.shN_ada_mask(param, step)
        return param, None

    def shN_gradient_threshold(self, param: torch.nn.Parameter, step: int) -> None:
        """reference simulation.py:327-348 (the "gradient" strategy; edits ``param.grad`` in place)."""
        _shN_gradient_threshold(param.data, param.grad)


class STGCompressionSimulation(_SimulationBase):
    """Dynamic-scene hooks (spacetime gaussians): explicit ``quantization_sim_type``
    ("round" in the dyngs preset), 17 floats per splat."""

    def __init__(self, quantization_sim_type: Optional[Literal["round", "noise", "vq"]] = None,
                 entropy_model_enable: bool = False, entropy_steps: Optional[Dict[str, int]] = None, device=None,
                 ada_mask_opt: bool = False, ada_mask_step: int = 10_000, **kwargs) -> None:
        self.quantization_sim_type = quantization_sim_type
        self.q_type = quantization_sim_type
        self.entropy_steps = entropy_steps
        self.device = device
        self.entropy_model_option = {"means": False, "scales": True, "quats": True, "opacities": False, "colors": True,
                                     "features_dir": True, "features_time": True}
        self.simulation_option = {
            "means": False, "scales": True, "quats": True, "opacities": True, "trbf_center": False,
            "trbf_scale": False, "motion": False, "omega": False, "colors": True, "features_dir": True,
            "features_time": True,
        }
        self.q_bitwidth = {
            "means": None, "scales": 8, "quats": 8, "opacities": 8, "trbf_center": None, "trbf_scale": None,
            "motion": None, "omega": None, "colors": 8, "features_dir": 8, "features_time": 8,
        }
        self.bds = {
            "means": None, "scales": [-10, 2], "quats": [-1, 1], "opacities": [-7, 7], "trbf_center": None,
            "trbf_scale": None, "motion": None, "omega": None, "colors": [-7.5, 7.5], "features_dir": [-10, 10],
            "features_time": [-10, 10],
        }
        self._setup_entropy(entropy_model_enable, "factorized_model", {  # reference simulation.py:587-596
            "means": None, "scales": dict(channel=3), "quats": dict(channel=4), "opacities": None,
            "colors": dict(channel=3, filters=(3, 3)), "features_dir": dict(channel=3, filters=(3, 3)),
            "features_time": dict(channel=3, filters=(3, 3))})
        # the dynamic trainer constructs the mask and its optimizer (simulation.py:611-620) although its splats carry no shN and
        # no simulate function consumes it: the attributes exist, as in the reference
        self._setup_ada_mask(ada_mask_opt, ada_mask_step, "learnable", kwargs)

    def simulate_compression_scales(self, param, step, *_): return self._quantize("scales", param, step)
    def simulate_compression_quats(self, param, step, *_): return self._quantize("quats", param, step)

    def simulate_compression_opacities(self, param, step, *_):
        return self._quantize("opacities", param, step, lambda v: v.unsqueeze(1))

    def simulate_compression_colors(self, param, step, *_): return self._quantize("colors", param, step)
    def simulate_compression_features_dir(self, param, step, *_): return self._quantize("features_dir", param, step)
    def simulate_compression_features_time(self, param, step, *_): return self._quantize("features_time", param, step)
