"""Per-attribute compression-simulation hooks (quantizer part).

Mirrors ``CompressionSimulation`` (reference simulation.py:14-348) and
``STGCompressionSimulation`` (508-780): which attributes are fake-quantized, their bounds
and bit widths, and the ``simulate_compression(splats, step) -> (new_splats, esti_bits)``
contract.  Entropy models (factorized prior / hash-grid Gaussian model) are outside this
hot path: constructing with ``entropy_model_enable=True`` raises NotImplementedError.
"""
from __future__ import annotations

from typing import Dict, Optional, Tuple

import torch
from torch import Tensor
from typing_extensions import Literal

from .ops import fake_quantize_ste


class _SimulationBase:
    simulation_option: Dict[str, bool]
    q_bitwidth: Dict[str, Optional[int]]
    bds: Dict[str, Optional[list]]
    q_type: Optional[str]

    def _check_entropy(self, enable: bool) -> None:
        if enable:
            raise NotImplementedError(
                "entropy models (bits estimators) are not part of the MI355X hot path yet "
                "(SURVEY.md section 8f rank 1); construct with entropy_model_enable=False"
            )

    def _quantize(self, name: str, param: Tensor) -> Tuple[Tensor, None]:
        lo, hi = self.bds[name]
        # both sides of the reference's `step < 10_000` branch select 8 bits
        # (simulation.py:242-245): the schedule is a no-op and q_bitwidth[name] is used.
        bits = self.q_bitwidth[name]
        if self.q_type is None:
            out = fake_quantize_ste(param, lo, hi, bits)  # default q_type="noise"
        else:
            out = fake_quantize_ste(param, lo, hi, bits, self.q_type)
        return out["output_value"], None

    def simulate_compression(self, splats: Dict[str, Tensor], step: int):
        """Returns (new_splats, esti_bits_dict); un-simulated attributes come back as ``p + 0.``"""
        new_splats, esti_bits = {}, {}
        for name in splats.keys():
            if self.simulation_option[name]:
                fn = getattr(self, f"simulate_compression_{name}", None)
                if fn is None:
                    raise NotImplementedError(f"no simulate function for {name}")
                new_splats[name], esti_bits[name] = fn(splats[name], step)
            else:
                new_splats[name] = splats[name] + 0.0
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
        self._check_entropy(entropy_model_enable)
        if ada_mask_opt:
            raise NotImplementedError("learnable shN mask (ada_mask_opt) is outside the hot path")
        self.entropy_model_enable = False
        self.entropy_model_type = entropy_model_type
        self.entropy_steps = entropy_steps
        self.device = device
        self.q_type = None
        self.simulation_option = {"means": False, "scales": True, "quats": True, "opacities": True, "sh0": True,
                                  "shN": True}
        self.q_bitwidth = {"means": None, "scales": 8, "quats": 8, "opacities": 8, "sh0": 8, "shN": None}
        self.bds = {"means": None, "scales": [-10, 2], "quats": [-1, 1], "opacities": [-15, 15], "sh0": [-2, 4],
                    "shN": None}

    def simulate_compression_scales(self, param, step): return self._quantize("scales", param)
    def simulate_compression_quats(self, param, step): return self._quantize("quats", param)
    def simulate_compression_opacities(self, param, step): return self._quantize("opacities", param)
    def simulate_compression_sh0(self, param, step): return self._quantize("sh0", param)

    def simulate_compression_shN(self, param, step):
        return param, None  # reference simulation.py:319-324 without the optional mask


class STGCompressionSimulation(_SimulationBase):
    """Dynamic-scene hooks (spacetime gaussians): explicit ``quantization_sim_type``
    ("round" in the dyngs preset), 17 floats per splat."""

    def __init__(self, quantization_sim_type: Optional[Literal["round", "noise", "vq"]] = None,
                 entropy_model_enable: bool = False, entropy_steps: Optional[Dict[str, int]] = None, device=None,
                 ada_mask_opt: bool = False, ada_mask_step: int = 10_000, **kwargs) -> None:
        self._check_entropy(entropy_model_enable)
        self.quantization_sim_type = quantization_sim_type
        self.q_type = quantization_sim_type
        self.entropy_model_enable = False
        self.entropy_steps = entropy_steps
        self.device = device
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

    def _quantize(self, name, param):
        lo, hi = self.bds[name]
        out = fake_quantize_ste(param, lo, hi, self.q_bitwidth[name], self.q_type)
        return out["output_value"], None

    def simulate_compression_scales(self, param, step): return self._quantize("scales", param)
    def simulate_compression_quats(self, param, step): return self._quantize("quats", param)
    def simulate_compression_opacities(self, param, step): return self._quantize("opacities", param)
    def simulate_compression_colors(self, param, step): return self._quantize("colors", param)
    def simulate_compression_features_dir(self, param, step): return self._quantize("features_dir", param)
    def simulate_compression_features_time(self, param, step): return self._quantize("features_time", param)
