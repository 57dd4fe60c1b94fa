"""Train-time compression simulation: per-splat quantize/dequantize STE hooks.

Counterpart of the quantizer part of the reference's ``gsplat/compression_simulation``
(ops.py + the quantization hooks of simulation.py).  The learned entropy models of the
reference are NOT part of this hot path (SURVEY.md section 8f, rank 1 "next").
"""
from .ops import STE, fake_quantize_ste
from .simulation import CompressionSimulation, STGCompressionSimulation

__all__ = ["STE", "fake_quantize_ste", "CompressionSimulation", "STGCompressionSimulation"]
