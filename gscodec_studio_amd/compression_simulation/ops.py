"""Quantize / dequantize straight-through estimators on the HIP path.

Same functions as the reference's ``gsplat/compression_simulation/ops.py`` (39-75):

* ``fake_quantize_ste(x, lo, hi, bitwidth=8, q_type="noise")`` -> ``{"output_value", "q_step"}``
    - "noise": ``clamp(x, lo, hi) + U(-0.5, 0.5) * q_step``; the gradient passes where
      ``lo <= x <= hi``.  The noise tensor is drawn by ``torch.empty_like(x).uniform_`` from
      the device's default generator exactly as the reference does, so the RNG stream is
      unchanged; clamp + scale + add run as ONE HIP kernel instead of four torch kernels.
    - "round": ``STE.apply`` -- clamps the PARAMETER IN PLACE (reference ops.py:63 mutates
      its input, kept on purpose), rounds half-to-even on the [0, 2^b-1] grid, identity
      gradient everywhere (including clamped elements).
* any other ``q_type`` (e.g. "vq", accepted by the reference's config type) raises the same
  ``UnboundLocalError`` the reference raises.
"""
from __future__ import annotations

import ctypes
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
from torch import Tensor

from .. import _backend as B
from .._wrapper import _device_of, _require_gpu, _stream


def _f32(v: float) -> float:
    """Round a Python double to the nearest fp32 (what torch does with a wrapped scalar)."""
    return float(np.float32(v))


_ACTS = {None: 0, "exp": 1, "sigmoid": 2}  # GS_ACT_* of include/gsplat_hip.h


class _NoiseQuant(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x: Tensor, noise: Tensor, lo: float, hi: float, q_step: float, act: int = 0) -> Tensor:
        _require_gpu(x, "fake_quantize_ste")
        if x.dtype != torch.float32:
            raise RuntimeError(f"fake_quantize_ste: expected float32, got {x.dtype}")
        xc = x.contiguous()
        out = torch.empty_like(xc)
        with _device_of(xc):
            B.call("gs_quantize_noise_fwd", xc.numel(), B.ptr(xc), B.ptr(noise), _f32(lo), _f32(hi), _f32(q_step), act,
                   B.ptr(out), _stream(xc))
        ctx.save_for_backward(xc, out if act else None)
        ctx.bounds, ctx.act = (_f32(lo), _f32(hi)), act
        return out.view(x.shape)

    @staticmethod
    def backward(ctx, v_out: Tensor):
        xc, out = ctx.saved_tensors
        lo, hi = ctx.bounds
        v_out = v_out.contiguous()
        v_x = torch.empty_like(xc)
        with _device_of(xc):
            B.call("gs_quantize_noise_bwd", xc.numel(), B.ptr(xc), B.ptr(v_out), lo, hi, ctx.act, B.ptr(out), B.ptr(v_x), _stream(xc))
        return v_x.view(v_out.shape), None, None, None, None, None


class STE(torch.autograd.Function):
    """Round-to-grid straight-through estimator (reference ops.py:57-75).

    ``STE.apply(input, bitdepth=8, min=-1, max=1)``.  ``input`` is clamped IN PLACE.
    """

    @staticmethod
    def forward(ctx, input: Tensor, bitdepth: int = 8, min: float = -1, max: float = 1, act: int = 0) -> Tensor:
        _require_gpu(input, "STE")
        if input.dtype != torch.float32:
            raise RuntimeError(f"STE: expected float32, got {input.dtype}")
        if not input.is_contiguous():
            raise RuntimeError("STE: input must be contiguous (it is clamped in place)")
        out = torch.empty_like(input)
        rng = _f32(max - min)  # python arithmetic first, then fp32, as torch does
        qn = _f32(1 / (2**bitdepth - 1))
        with _device_of(input):
            B.call("gs_quantize_round_fwd", input.numel(), B.ptr(input), _f32(min), _f32(max), rng, qn, act, B.ptr(out),
                   _stream(input))
        ctx.act = act
        if act:
            ctx.save_for_backward(out)
        return out

    @staticmethod
    def backward(ctx, grad_output: Tensor):
        if not ctx.act:
            return grad_output, None, None, None, None  # identity, everywhere (ops.py:73-75)
        (out,) = ctx.saved_tensors
        g = grad_output.contiguous()
        v_x = torch.empty_like(out)
        with _device_of(out):
            B.call("gs_quantize_round_bwd", out.numel(), B.ptr(g), ctx.act, B.ptr(out), B.ptr(v_x), _stream(out))
        return v_x, None, None, None, None


class _QuantDesc(ctypes.Structure):  # gs_quant_desc of include/gsplat_hip.h
    _fields_ = [("n", ctypes.c_uint64), ("x", ctypes.c_void_p), ("out", ctypes.c_void_p), ("v_out", ctypes.c_void_p),
                ("v_x", ctypes.c_void_p), ("lo", ctypes.c_float), ("hi", ctypes.c_float), ("q_step", ctypes.c_float),
                ("activation", ctypes.c_int32), ("philox_offset", ctypes.c_uint64)]


QUANT_MULTI_MAX = 8
_DESC_CHECKED = [False]


def check_desc_layout() -> None:
    """``_QuantDesc`` against the library's ``sizeof`` / ``offsetof`` of ``gs_quant_desc``."""
    want = (ctypes.c_uint64 * 16)()
    m = int(B.query("gs_quant_desc_layout", want, 16))
    mine = [ctypes.sizeof(_QuantDesc)] + [getattr(_QuantDesc, f).offset for f in ("n", "x", "out", "v_out", "v_x", "lo", "q_step", "activation",
                                                                                  "philox_offset")]
    if m != len(mine) or list(want[:m]) != mine:
        raise ImportError(f"gs_quant_desc: the ctypes mirror in ops.py does not match the library's struct layout ({list(want[:m])} vs {mine})")
    _DESC_CHECKED[0] = True
_GRID_CAP: Dict[int, int] = {}


def _grid_cap(device: torch.device) -> int:
    """What torch caps the grid of its random kernels at: CUs * (max threads per CU / 256)."""
    i = device.index if device.index is not None else torch.cuda.current_device()
    if i not in _GRID_CAP:
        p = torch.cuda.get_device_properties(i)
        _GRID_CAP[i] = p.multi_processor_count * (p.max_threads_per_multi_processor // 256)
    return _GRID_CAP[i]


class _NoiseQuantMulti(torch.autograd.Function):
    """``fake_quantize_ste(x_i, lo_i, hi_i, bits_i, "noise")`` for several tensors in ONE launch each way, the noise generated in
    the kernel from the device's default generator exactly as the tensors' ``uniform_`` calls would have drawn it, in order
    (the generator is advanced by the same amounts): bit-identical outputs, same RNG stream afterwards."""

    @staticmethod
    def forward(ctx, specs: Sequence[Tuple[float, float, float, int]], *xs: Tensor):
        dev = xs[0].device
        if not _DESC_CHECKED[0]:
            check_desc_layout()
        for x in xs:
            _require_gpu(x, "fake_quantize_ste")
            if x.dtype != torch.float32 or x.device != dev:
                raise RuntimeError("fake_quantize_ste (multi): float32 tensors on one device")
        xc = [x.contiguous() for x in xs]
        outs = [torch.empty_like(x) for x in xc]
        gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
        seed, off = gen.initial_seed(), gen.get_offset()
        cap = _grid_cap(dev)
        descs = (_QuantDesc * len(xc))()
        for d, x, o, (lo, hi, q_step, act) in zip(descs, xc, outs, specs):
            d.n, d.x, d.out, d.v_out, d.v_x = x.numel(), B.ptr(x), B.ptr(o), None, None
            d.lo, d.hi, d.q_step, d.activation, d.philox_offset = _f32(lo), _f32(hi), _f32(q_step), act, off
            off += int(B.query("gs_quantize_philox_advance", x.numel(), cap))
        gen.set_offset(off)
        with _device_of(xc[0]):
            B.call("gs_quantize_noise_multi_fwd", len(xc), ctypes.addressof(descs), seed, cap, _stream(xc[0]))
        ctx.specs = [(_f32(lo), _f32(hi), act) for lo, hi, _, act in specs]
        ctx.save_for_backward(*xc, *[o if sp[3] else None for o, sp in zip(outs, specs)])
        ctx.set_materialize_grads(False)
        return tuple(o.view(x.shape) for o, x in zip(outs, xs))

    @staticmethod
    def backward(ctx, *v_outs):
        k = len(ctx.specs)
        xc, outs = ctx.saved_tensors[:k], ctx.saved_tensors[k:]
        descs = (_QuantDesc * k)()
        grads: List[Optional[Tensor]] = []
        live = []
        for i, (x, o, v, (lo, hi, act)) in enumerate(zip(xc, outs, v_outs, ctx.specs)):
            if v is None or not ctx.needs_input_grad[1 + i]:
                grads.append(None)
                descs[i].n = 0
                continue
            g = v.contiguous()
            vx = torch.empty_like(x)
            live.append(g)
            d = descs[i]
            d.n, d.x, d.out, d.v_out, d.v_x = x.numel(), B.ptr(x), B.ptr(o), B.ptr(g), B.ptr(vx)
            d.lo, d.hi, d.q_step, d.activation, d.philox_offset = lo, hi, 0.0, act, 0
            grads.append(vx.view(v.shape))
        if live:
            with _device_of(xc[0]):
                B.call("gs_quantize_noise_multi_bwd", k, ctypes.addressof(descs), _stream(xc[0]))
        return (None, *grads)


_SELFCHECK: Dict[int, bool] = {}


def multi_selfcheck(device: torch.device) -> bool:
    """Once per device: does the in-kernel noise of ``_NoiseQuantMulti`` still equal torch's ``uniform_`` stream?  The kernel
    restates torch internals by hand (grid cap CUs x max-threads / 256, unroll 4, the Philox offset advance, rocrand's
    ``v 2^-32 + 2^-32`` conversion): a torch or ROCm change would silently break the bit-identity with the tensor-by-tensor hooks.
    Two tensors (one spanning several grid strides, one ragged) are quantized both ways from the same generator state; the
    values AND the generator offset afterwards must agree.  The caller's RNG state is restored.  False (+ a warning) disables
    the one-launch path for the process -- the per-tensor calls are the reference's own sequence."""
    i = device.index if device.index is not None else torch.cuda.current_device()
    if i in _SELFCHECK:
        return _SELFCHECK[i]
    gen = torch.cuda.default_generators[i]
    state = gen.get_state()
    ok = False
    try:
        with torch.no_grad():
            xs = [torch.linspace(-3.0, 3.0, 300_007, device=device), torch.linspace(-1.0, 1.0, 1_001, device=device)]
            bounds, bits = [(-2.0, 2.0), (-1.0, 1.0)], [8, 8]
            gen.manual_seed(0x5EED)
            multi = fake_quantize_noise_multi(xs, bounds, bits)
            off_multi = gen.get_offset()
            gen.manual_seed(0x5EED)
            single = [fake_quantize_ste(x, lo, hi, b, "noise") for x, (lo, hi), b in zip(xs, bounds, bits)]
            off_single = gen.get_offset()
            ok = off_multi == off_single and all(torch.equal(m["output_value"], s_["output_value"]) for m, s_ in zip(multi, single))
    except Exception as e:  # (a missing symbol, an unexpected generator API: the per-tensor path stays)
        ok = False
        import warnings

        warnings.warn(f"gscodec_studio_amd: multi-tensor quantizer self-check failed to run ({type(e).__name__}: {e})")
    finally:
        gen.set_state(state)
    if not ok:
        import warnings

        warnings.warn("gscodec_studio_amd: the in-kernel noise of the multi-tensor quantizer no longer reproduces torch's uniform_ "
                      "stream on this torch / ROCm build; falling back to the per-tensor hooks (GS_QUANT_MULTI=0 behaviour)")
    _SELFCHECK[i] = ok
    return ok


def fake_quantize_noise_multi(inputs: Sequence[Tensor], bounds: Sequence[Tuple[float, float]], bitwidths: Sequence[int],
                              activations: Optional[Sequence[Optional[str]]] = None) -> List[Dict[str, object]]:
    """``[fake_quantize_ste(x, lo, hi, bits, "noise", activation) for ...]`` -- same outputs, same RNG stream -- in one launch
    (up to QUANT_MULTI_MAX tensors; not in the reference, whose hooks run tensor by tensor: simulation.py:206-324)."""
    assert 1 <= len(inputs) <= QUANT_MULTI_MAX and len(inputs) == len(bounds) == len(bitwidths)
    acts = list(activations) if activations is not None else [None] * len(inputs)
    q_steps = [(hi - lo) / (2**b - 1) for (lo, hi), b in zip(bounds, bitwidths)]
    specs = [(lo, hi, q, _ACTS[a]) for (lo, hi), q, a in zip(bounds, q_steps, acts)]
    outs = _NoiseQuantMulti.apply(specs, *inputs)
    return [{"output_value": o, "q_step": q} for o, q in zip(outs, q_steps)]


class _RoundQuantMulti(torch.autograd.Function):
    """``STE.apply(x_i, bits_i, lo_i, hi_i[, act_i])`` for several tensors in ONE launch (``gs_quantize_round_multi_fwd``): every ``x_i`` is
    clamped IN PLACE like ``STE``'s input, the outputs are the grid values (activated where asked); backward = identity, times the
    activation's derivative where one was fused (one launch for those)."""

    @staticmethod
    def forward(ctx, specs: Sequence[Tuple[float, float, int, int]], *xs: Tensor):
        dev = xs[0].device
        if not _DESC_CHECKED[0]:
            check_desc_layout()
        for x in xs:
            _require_gpu(x, "STE")
            if x.dtype != torch.float32 or x.device != dev or not x.is_contiguous():
                raise RuntimeError("STE (multi): contiguous float32 tensors on one device (they are clamped in place)")
        outs = [torch.empty_like(x) for x in xs]
        k = len(xs)
        descs = (_QuantDesc * k)()
        ranges, qns = (ctypes.c_float * k)(), (ctypes.c_float * k)()
        for i, (d, x, o, (lo, hi, bits, act)) in enumerate(zip(descs, xs, outs, specs)):
            d.n, d.x, d.out, d.v_out, d.v_x = x.numel(), B.ptr(x), B.ptr(o), None, B.ptr(x)
            d.lo, d.hi, d.q_step, d.activation, d.philox_offset = _f32(lo), _f32(hi), 0.0, act, 0
            ranges[i], qns[i] = _f32(hi - lo), _f32(1 / (2**bits - 1))  # python arithmetic first, then fp32, as torch does
        with _device_of(xs[0]):
            B.call("gs_quantize_round_multi_fwd", k, ctypes.addressof(descs), ctypes.addressof(ranges), ctypes.addressof(qns), _stream(xs[0]))
        ctx.acts = [sp[3] for sp in specs]
        ctx.save_for_backward(*[o if a else None for o, a in zip(outs, ctx.acts)])
        ctx.set_materialize_grads(False)
        return tuple(outs)

    @staticmethod
    def backward(ctx, *v_outs):
        outs = ctx.saved_tensors
        k = len(ctx.acts)
        grads: List[Optional[Tensor]] = [None] * k
        descs = (_QuantDesc * k)()
        live = []
        for i, (o, v, act) in enumerate(zip(outs, v_outs, ctx.acts)):
            descs[i].n = 0
            if v is None or not ctx.needs_input_grad[1 + i]:
                continue
            if not act:
                grads[i] = v  # identity, everywhere (ops.py:73-75)
                continue
            g = v.contiguous()
            vx = torch.empty_like(o)
            live.append(g)
            d = descs[i]
            d.n, d.x, d.out, d.v_out, d.v_x = o.numel(), B.ptr(o), B.ptr(o), B.ptr(g), B.ptr(vx)
            d.lo, d.hi, d.q_step, d.activation, d.philox_offset = 0.0, 0.0, 0.0, act, 0
            grads[i] = vx
        if live:
            with _device_of(live[0]):
                B.call("gs_quantize_round_multi_bwd", k, ctypes.addressof(descs), _stream(live[0]))
        return (None, *grads)


def fake_quantize_round_multi(inputs: Sequence[Tensor], bounds: Sequence[Tuple[float, float]], bitwidths: Sequence[int],
                              activations: Optional[Sequence[Optional[str]]] = None) -> List[Dict[str, object]]:
    """``[fake_quantize_ste(x, lo, hi, bits, "round", activation) for ...]`` -- same outputs, same in-place clamp of every input -- in
    one launch (up to QUANT_MULTI_MAX contiguous float32 tensors; not in the reference, whose hooks run tensor by tensor)."""
    assert 1 <= len(inputs) <= QUANT_MULTI_MAX and len(inputs) == len(bounds) == len(bitwidths)
    acts = list(activations) if activations is not None else [None] * len(inputs)
    specs = [(lo, hi, b, _ACTS[a]) for (lo, hi), b, a in zip(bounds, bitwidths, acts)]
    outs = _RoundQuantMulti.apply(specs, *inputs)
    return [{"output_value": o, "q_step": (hi - lo) / (2**b - 1)} for o, (lo, hi), b in zip(outs, bounds, bitwidths)]


def fake_quantize_ste(input: Tensor, lower_bd: float, upper_bd: float, bitwidth: int = 8,
                      q_type: str = "noise", activation: str = None) -> Dict[str, object]:
    """``activation`` (opt-in, not in the reference): "exp" / "sigmoid" -- the activation the trainer applies to the hooked
    value right afterwards (reference examples/simple_trainer.py:779-786), evaluated in the quantizer's own pass; the
    returned ``output_value`` is then ALREADY activated.  The noise is drawn exactly as without it."""
    q_step = (upper_bd - lower_bd) / (2**bitwidth - 1)
    act = _ACTS[activation]

    if q_type == "round":
        output_value = STE.apply(input, bitwidth, lower_bd, upper_bd, act)
    elif q_type == "noise":
        noise = torch.empty_like(input, memory_format=torch.contiguous_format).uniform_(-0.5, 0.5)
        output_value = _NoiseQuant.apply(input, noise, lower_bd, upper_bd, q_step, act)

    out_dict = {
        "output_value": output_value,  # UnboundLocalError for unknown q_type, like the reference
        "q_step": q_step,
    }
    return out_dict
