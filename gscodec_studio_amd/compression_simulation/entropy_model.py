"""Factorized-prior bits estimator behind the quantize hooks.

``Entropy_factorized_optimized_refactor`` keeps the reference's constructor, parameter names
(``_matrices``, ``_bias``, ``_factor`` ParameterLists, same shapes and initialisation) and
``forward(x, Q) -> bits [N, C]`` contract (gsplat/compression_simulation/entropy_model.py:84-254),
so state dicts and optimizers are interchangeable; the ~30-kernel torch graph is replaced by one
fused HIP kernel each way (csrc/entropy.hip, ``gs_entropy_factorized_fwd/bwd``).  There is no torch
fallback: CPU tensors raise.  The hash-grid Gaussian model (``Entropy_gaussian``) needs the
reference's CUDA-only ``_gridencoder`` extension and is not provided.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn
from torch import Tensor

from .. import _backend as B


_REPLICAS = 32  # copies of the parameter-gradient buffer the workgroups spread their atomics over


def _stream(t: Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


class _FactorizedBits(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, half_q: Tensor, params: Tensor, layers: int, width: int, bound: float) -> Tensor:
        if not x.is_cuda:
            raise RuntimeError("factorized bits estimator: the HIP path needs device tensors (no CPU fallback)")
        x = x.contiguous()
        bits = torch.empty_like(x)
        with torch.cuda.device(x.device):
            B.call("gs_entropy_factorized_fwd", x.shape[0], x.shape[1], layers, width, B.ptr(x), B.ptr(half_q),
                   B.ptr(params), float(bound), B.ptr(bits), _stream(x))
        ctx.save_for_backward(x, half_q, params)
        ctx.cfg = (layers, width, float(bound))
        return bits

    @staticmethod
    def backward(ctx, v_bits: Tensor):
        x, half_q, params = ctx.saved_tensors
        layers, width, bound = ctx.cfg
        v_bits = v_bits.contiguous()
        v_x = torch.empty_like(x)
        v_params = torch.zeros((_REPLICAS,) + tuple(params.shape), dtype=params.dtype, device=params.device)
        with torch.cuda.device(x.device):
            B.call("gs_entropy_factorized_bwd", x.shape[0], x.shape[1], layers, width, B.ptr(x), B.ptr(half_q),
                   B.ptr(params), bound, B.ptr(v_bits), B.ptr(v_x), B.ptr(v_params), _REPLICAS, _stream(x))
        return ((v_x if ctx.needs_input_grad[0] else None), None,
                (v_params.sum(0) if ctx.needs_input_grad[2] else None), None, None, None)


class LowerBound(nn.Module):
    """``max(x, bound)`` whose gradient passes when x >= bound or the gradient pushes x up
    (reference entropy_model.py:347-394).  Kept for API parity; the fused kernel applies it itself."""

    def __init__(self, bound: float):
        super().__init__()
        self.register_buffer("bound", torch.Tensor([float(bound)]))

    def forward(self, x: Tensor) -> Tensor:
        return _LowerBoundFn.apply(x, self.bound)


class _LowerBoundFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, bound):
        ctx.save_for_backward(x, bound)
        return torch.max(x, bound)

    @staticmethod
    def backward(ctx, g):
        x, bound = ctx.saved_tensors
        return ((x >= bound) | (g < 0)) * g, None


class Entropy_factorized_optimized_refactor(nn.Module):
    """bits[n, c] = -log2 P(x[n, c] - Q/2 < X <= x[n, c] + Q/2) under a learned factorized density
    (Balle et al. 2018 style cumulative MLP).  Same parameters as the reference module; see
    csrc/entropy.hip for the arithmetic and the reproduced 32-way reshape quirk."""

    def __init__(self, channel=32, init_scale=10, filters=(3, 3, 3), likelihood_bound=1e-6, tail_mass=1e-9,
                 optimize_integer_offset=True, Q=1):
        super().__init__()
        self.channel = int(channel)
        self.filters = tuple(int(t) for t in filters)
        self.init_scale = float(init_scale)
        self.tail_mass = float(tail_mass)
        self.optimize_integer_offset = bool(optimize_integer_offset)
        self.Q = Q
        if not 0 < self.tail_mass < 1:
            raise ValueError("`tail_mass` must be between 0 and 1")
        if len(self.filters) < 1 or len(set(self.filters)) != 1 or not (1 <= self.filters[0] <= 4) or len(self.filters) > 4:
            raise NotImplementedError(
                f"filters={self.filters}: the HIP kernel supports 1..4 hidden layers of one common width 1..4 "
                "(the reference uses (3, 3) and (3, 3, 3))")
        if not 1 <= self.channel <= 32:
            raise NotImplementedError("channel must be in 1..32")
        widths = (1,) + self.filters + (1,)
        scale = self.init_scale ** (1.0 / (len(self.filters) + 1))
        self._matrices = nn.ParameterList([])
        self._bias = nn.ParameterList([])
        self._factor = nn.ParameterList([])
        for i in range(len(self.filters) + 1):  # initialisation as reference entropy_model.py:107-124
            init = np.log(np.expm1(1.0 / scale / widths[i + 1]))
            # the reference also binds every new Parameter to the attributes `matrix` / `bias` / `factor`
            # (entropy_model.py:112-126), so its state dicts carry those three keys (aliases of the LAST layer's tensors)
            # and named_parameters() lists the last layer under them: same registration order here, so reference
            # checkpoints load with strict=True and per-parameter optimizer groups get the same names
            self.matrix = nn.Parameter(torch.full((self.channel, widths[i + 1], widths[i]), float(init)))
            self._matrices.append(self.matrix)
            noise = np.random.uniform(-0.5, 0.5, (self.channel, widths[i + 1], 1))
            self.bias = nn.Parameter(torch.from_numpy(noise).float())
            self._bias.append(self.bias)
            if i < len(self.filters):
                self.factor = nn.Parameter(torch.zeros(self.channel, widths[i + 1], 1))
                self._factor.append(self.factor)
        self.register_buffer("filters_len", torch.tensor(len(self.filters)))
        self.register_buffer("factor_len", torch.tensor(len(self._factor)))
        self.likelihood_bound = float(likelihood_bound)
        self.likelihood_lower_bound = LowerBound(likelihood_bound)
        expect = B.query("gs_entropy_factorized_params_per_channel", len(self.filters), self.filters[0])
        self._n_params = self._packed_width()
        if expect != self._n_params:
            raise RuntimeError(f"parameter layout mismatch with libgsplat_hip ({expect} vs {self._n_params})")

    def _packed_width(self) -> int:
        w, L = self.filters[0], len(self.filters)
        return 3 * w + (L - 1) * (w * w + 2 * w) + w + 1

    def packed_parameters(self) -> Tensor:
        """[channel, P]: per layer [matrix row-major | bias | factor] (differentiable view of the ParameterLists)."""
        parts = []
        for i in range(len(self._matrices)):
            parts.append(self._matrices[i].reshape(self.channel, -1))
            parts.append(self._bias[i].reshape(self.channel, -1))
            if i < len(self._factor):
                parts.append(self._factor[i].reshape(self.channel, -1))
        return torch.cat(parts, dim=1).contiguous()

    def _half_q(self, x: Tensor, Q) -> Tensor:
        if Q is None:
            Q = self.Q
        if isinstance(Q, Tensor):
            q = Q.detach().to(device=x.device, dtype=torch.float32).reshape(-1)
            if q.numel() == 1:
                q = q.expand(self.channel)
            elif q.numel() != self.channel:
                raise ValueError(f"Q must have 1 or {self.channel} elements, got {q.numel()}")
            return (0.5 * q).contiguous()
        return torch.full((self.channel,), 0.5 * float(Q), dtype=torch.float32, device=x.device)

    def forward(self, x: Tensor, Q=None, **kwargs) -> Tensor:
        assert x.dim() == 2 and x.shape[1] == self.channel, f"x must be [N, {self.channel}], got {tuple(x.shape)}"
        if x.dtype != torch.float32:
            x = x.float()
        return _FactorizedBits.apply(x, self._half_q(x, Q), self.packed_parameters(), len(self.filters), self.filters[0],
                                     self.likelihood_bound)

    def get_likelihood(self, x: Tensor, Q=None, **kwargs) -> Tensor:
        """The bounded likelihood (reference entropy_model.py:256-305) = 2^-bits."""
        return torch.exp2(-self.forward(x, Q))
