"""Train-time compression simulation: per-splat quantize/dequantize STE hooks.

Counterpart of the quantizer part of the reference's ``gsplat/compression_simulation``
(ops.py + the quantization hooks of simulation.py) and of its factorized-prior bits estimator
(entropy_model.py) and of the learnable shN mask (ada_mask.py).  The hash-grid Gaussian
entropy model is not built.
"""
from .ada_mask import AnnealingMask
from .entropy_model import Entropy_factorized_optimized_refactor, LowerBound
from .ops import STE, fake_quantize_ste
from .simulation import CompressionSimulation, STGCompressionSimulation

__all__ = ["AnnealingMask", "STE", "fake_quantize_ste", "CompressionSimulation", "STGCompressionSimulation",
           "Entropy_factorized_optimized_refactor", "LowerBound"]
