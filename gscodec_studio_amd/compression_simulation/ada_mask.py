"""Learnable per-splat mask on the higher SH bands ("shN adaptive mask"), HIP path.

Same class as the reference's ``gsplat/compression_simulation/ada_mask.py:6-62`` (``AnnealingMask``: constructor
arguments, ``mask_logits`` parameter of shape ``input_shape`` initialised to 1, the exponential temperature schedule,
``forward(x, current_step)``, ``get_binary_mask``, ``get_sparsity_loss``, ``get_mask_ratio``, ``current_iter``), with the
elementwise torch chains replaced by the kernels of ``csrc/ada_mask.hip``:

* ``forward``: ``x * sigmoid(mask_logits / T(step))`` in training mode, ``x * (sigmoid(mask_logits) >= 0.5)`` in eval mode
  -- one streaming pass (``gs_shn_mask_fwd``); the backward (``gs_shn_mask_bwd``) returns ``v_x`` and the per-splat logit
  gradient in the same pass, reduced without atomics;
* ``get_sparsity_loss``: the mean of the soft mask is one deterministic reduction (``gs_mask_sum``) with a hand-written
  backward (``gs_mask_mean_bwd``); the scalar L1 + binary-cross-entropy arithmetic on top of it stays in torch, as written in
  the reference (ada_mask.py:46-58);
* ``get_mask_ratio``: the same reduction in binary mode.

There is no CPU fallback: CPU tensors raise ``RuntimeError``.
"""
from __future__ import annotations

import math

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch import Tensor

from .. import _backend as B
from .._wrapper import _device_of, _require_gpu, _stream


def _f32(v: float) -> float:
    return float(np.float32(v))


class _ShNMask(torch.autograd.Function):
    """out[n, ...] = x[n, ...] * mask(logits[n]); x [N, ...] with any trailing shape, logits N values."""

    @staticmethod
    def forward(ctx, x: Tensor, logits: Tensor, temperature: float, binary: bool) -> Tensor:
        _require_gpu(x, "AnnealingMask")
        _require_gpu(logits, "AnnealingMask")
        if x.dtype != torch.float32 or logits.dtype != torch.float32:
            raise RuntimeError(f"AnnealingMask: expected float32, got {x.dtype} / {logits.dtype}")
        n = logits.numel()
        if x.dim() < 2 or x.shape[0] != n:
            # the reference multiplies x [N, K-1, 3] by mask_logits [cap_max, 1, 1]: broadcasting fails unless N == cap_max
            raise RuntimeError(f"The size of tensor a ({x.shape[0] if x.dim() else 0}) must match the size of tensor b ({n}) "
                               "at non-singleton dimension 0")
        row = x.numel() // n if n else 0
        if n and row < 2:
            raise RuntimeError("AnnealingMask: x must hold at least two values per splat")
        xc, lc = x.contiguous(), logits.contiguous()
        out = torch.empty_like(xc)
        with _device_of(xc):
            B.call("gs_shn_mask_fwd", n, row, B.ptr(xc), B.ptr(lc), _f32(temperature), int(binary), B.ptr(out), _stream(xc))
        ctx.save_for_backward(xc, lc)
        ctx.cfg = (n, row, _f32(temperature), int(binary), logits.shape)
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, v_out: Tensor):
        xc, lc = ctx.saved_tensors
        n, row, temperature, binary, lshape = ctx.cfg
        want_x, want_l = ctx.needs_input_grad[0], ctx.needs_input_grad[1] and not binary
        if n == 0 or not (want_x or want_l):
            return (torch.zeros_like(xc) if want_x else None), (torch.zeros(lshape, device=lc.device) if want_l else None), None, None
        g = v_out.contiguous()
        v_x = torch.empty_like(xc) if want_x else None
        v_l = torch.empty_like(lc) if want_l else None
        with _device_of(xc):
            B.call("gs_shn_mask_bwd", n, row, B.ptr(xc), B.ptr(lc), temperature, binary, B.ptr(g), B.ptr(v_x), B.ptr(v_l), _stream(xc))
        return (v_x.view(v_out.shape) if want_x else None), (v_l.view(lshape) if want_l else None), None, None


def _mask_sum(logits: Tensor, temperature: float, binary: bool, divisor: float) -> Tensor:
    lc = logits.contiguous()
    out = torch.empty((), device=lc.device, dtype=torch.float32)
    temp = torch.empty(B.query("gs_mask_sum_temp_bytes"), device=lc.device, dtype=torch.uint8)
    with _device_of(lc):
        B.call("gs_mask_sum", lc.numel(), B.ptr(lc), _f32(temperature), int(binary), _f32(divisor), B.ptr(temp), B.ptr(out), _stream(lc))
    return out


class _MaskMean(torch.autograd.Function):
    """mean(sigmoid(logits / T)) as a 0-dim tensor."""

    @staticmethod
    def forward(ctx, logits: Tensor, temperature: float) -> Tensor:
        _require_gpu(logits, "AnnealingMask.get_sparsity_loss")
        out = _mask_sum(logits, temperature, False, float(logits.numel()))
        ctx.save_for_backward(logits)
        ctx.temperature = _f32(temperature)
        return out

    @staticmethod
    def backward(ctx, v_mean: Tensor):
        (logits,) = ctx.saved_tensors
        lc = logits.contiguous()
        g = v_mean.contiguous().to(torch.float32)
        v_l = torch.empty_like(lc)
        with _device_of(lc):
            B.call("gs_mask_mean_bwd", lc.numel(), B.ptr(lc), ctx.temperature, B.ptr(g), _f32(float(lc.numel())), B.ptr(v_l), _stream(lc))
        return v_l.view(logits.shape), None


class MaskedShN:
    """``shN`` together with the mask that is to multiply it -- what ``AnnealingMask.fused`` (and ``simulate_compression(...,
    activate=True)``) hand to ``rasterization(colors=(sh0, MaskedShN))`` instead of the masked tensor: the projection pass then
    multiplies the coefficients as it loads them and its backward returns both gradients, so the 180 bytes per splat of masked
    coefficients (and of their gradient) are never written or read (opt-in, not in the reference; same values).  Any consumer
    that needs the tensor calls ``materialize()``."""

    __slots__ = ("shN", "mask_logits", "temperature", "binary")

    def __init__(self, shN: Tensor, mask_logits: Tensor, temperature: float, binary: bool):
        self.shN, self.mask_logits, self.temperature, self.binary = shN, mask_logits, float(temperature), bool(binary)

    @property
    def shape(self):
        return self.shN.shape

    def materialize(self) -> Tensor:
        return _ShNMask.apply(self.shN, self.mask_logits, self.temperature, self.binary)


class AnnealingMask(nn.Module):
    """reference ada_mask.py:6-62"""

    def __init__(self, input_shape, device, total_iters=30_000, start_temp=5.0, end_temp=0.1, annealing_start_iter=10_000,
                 target_sparsity=0.2):
        super().__init__()
        self.mask_logits = nn.Parameter(torch.zeros(input_shape, device=device) + 1)
        self.total_iters = total_iters
        self.start_temp = start_temp
        self.end_temp = end_temp
        self.annealing_start_iter = annealing_start_iter
        self.current_iter = 0
        self.target_sparsity = target_sparsity

    def get_temperature(self, current_step):
        if current_step < self.annealing_start_iter:
            return self.start_temp
        progress = (current_step - self.annealing_start_iter) / (self.total_iters - self.annealing_start_iter)
        progress = min(max(progress, 0), 1)
        return self.start_temp * math.exp(math.log(self.end_temp / self.start_temp) * progress)

    def forward(self, x, current_step):
        if self.training:
            self.current_iter = current_step
            return _ShNMask.apply(x, self.mask_logits, self.get_temperature(current_step), False)
        return _ShNMask.apply(x, self.mask_logits, 1.0, True)

    def fused(self, x, current_step) -> MaskedShN:
        """``forward`` without the multiplication: (x, logits, temperature, binary) for the renderer to apply (``MaskedShN``)."""
        if self.training:
            self.current_iter = current_step
            return MaskedShN(x, self.mask_logits, self.get_temperature(current_step), False)
        return MaskedShN(x, self.mask_logits, 1.0, True)

    @torch.no_grad()
    def get_binary_mask(self):
        """(sigmoid(mask_logits) >= 0.5).float(), shape of mask_logits (broadcasts against shN [N, K-1, 3])."""
        _require_gpu(self.mask_logits, "AnnealingMask.get_binary_mask")
        lc = self.mask_logits.detach().contiguous()
        out = torch.empty_like(lc)
        with _device_of(lc):
            B.call("gs_mask_values", lc.numel(), B.ptr(lc), 1.0, 1, B.ptr(out), _stream(lc))
        return out

    def get_sparsity_loss(self, lambda_l1=0.01, lambda_target=0.1):
        temperature = self.get_temperature(self.current_iter)
        mean_mask = _MaskMean.apply(self.mask_logits, temperature)  # torch.mean(mask), used twice in the reference
        l1_loss = lambda_l1 * mean_mask
        target = torch.tensor(self.target_sparsity).to(mean_mask.device)
        kl_loss = lambda_target * F.binary_cross_entropy(mean_mask, target)
        return l1_loss + kl_loss

    @torch.no_grad()
    def get_mask_ratio(self):
        _require_gpu(self.mask_logits, "AnnealingMask.get_mask_ratio")
        return _mask_sum(self.mask_logits.detach(), 1.0, True, float(self.mask_logits.shape[0]))


def shN_gradient_threshold(param: Tensor, grad: Tensor) -> None:
    """The "gradient" strategy (reference simulation.py:327-348): zero, IN PLACE, the gradient rows of splats whose shN row
    is all zero and whose gradient norm is below the threshold (2e-3 when fewer than 10 % of the rows are non-zero, else
    100) -- two launches, no host read-back (``gs_shn_grad_threshold``)."""
    _require_gpu(param, "shN_gradient_threshold")
    _require_gpu(grad, "shN_gradient_threshold")
    if not (param.is_contiguous() and grad.is_contiguous()) or param.shape != grad.shape or grad.dtype != torch.float32:
        raise RuntimeError("shN_gradient_threshold: param and its gradient must be contiguous float32 tensors of one shape")
    n = param.shape[0]
    if n == 0:
        return
    row = param.numel() // n
    flags = torch.empty(n, device=param.device, dtype=torch.uint8)
    count = torch.empty(1, device=param.device, dtype=torch.int64)
    with _device_of(param):
        B.call("gs_shn_grad_threshold", n, row, B.ptr(param), B.ptr(grad), B.ptr(flags), B.ptr(count), _stream(param))
