"""Dynamic (spacetime) gaussians: temporal slicing at one timestamp.

``temporal_slice`` is the elementwise chain that the reference's dynamic-scene trainer runs in front of
``rasterization()`` (examples/simple_trainer_dyngs.py:506-536; the viewer's copy:
examples/simple_viewer_dyn.py:84-101), as ONE HIP kernel each way (csrc/dynamic.hip) instead of ~25 torch
kernels.  Same inputs as the trainer's local variables: ``opacities`` and ``trbf_scale`` are the
ACTIVATED values (sigmoid / exp are applied by the caller, as in the trainer).  No torch fallback.

``DynamicSlice`` (round 6) hands the same slice to ``rasterization(dynamic=...)``, whose projection kernels then evaluate it in
their load phase (csrc/projection_dyn.hip) -- bit-identical to ``temporal_slice`` followed by ``rasterization``, without the
round trip of means_t / quats_t / opacity_t through HBM; opt-in on top, the trainer's activations (``raw``) and the round
quantizer hooks (``quantize``) ride in the same pass.  ``render_dynamic`` is the dynamic trainer's ``rasterize_splats`` in that form.
"""
from __future__ import annotations

import ctypes
from typing import Dict, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor

from . import _backend as B

TEMPORAL_VISIBILITY_THRESHOLD = 0.05  # simple_trainer_dyngs.py:526


class _TemporalSlice(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means, motion, quats, omega, opacities, trbf_center, trbf_scale, timestamp):
        if not means.is_cuda:
            raise RuntimeError("temporal_slice: the HIP path needs device tensors (no CPU fallback)")
        ins = [t.contiguous().float() for t in (means, motion, quats, omega, opacities, trbf_center, trbf_scale)]
        n = ins[0].shape[0]
        dev = ins[0].device
        means_t = torch.empty((n, 3), device=dev)
        quats_t = torch.empty((n, 4), device=dev)
        opacity_t = torch.empty((n,), device=dev)
        trbf = torch.empty((n,), device=dev)
        with torch.cuda.device(dev):
            B.call("gs_temporal_slice_fwd", n, *[B.ptr(t) for t in ins], float(timestamp), B.ptr(means_t), B.ptr(quats_t),
                   B.ptr(opacity_t), B.ptr(trbf), torch.cuda.current_stream(dev).cuda_stream)
        ctx.save_for_backward(*ins)
        ctx.timestamp = float(timestamp)
        ctx.set_materialize_grads(False)
        return means_t, quats_t, opacity_t, trbf

    @staticmethod
    def backward(ctx, v_means_t, v_quats_t, v_opacity_t, v_trbf):
        ins = ctx.saved_tensors
        n = ins[0].shape[0]
        dev = ins[0].device
        need = ctx.needs_input_grad
        outs = [torch.empty_like(t) if need[i] else None for i, t in enumerate(ins)]
        vin = [None if v is None else v.contiguous().float() for v in (v_means_t, v_quats_t, v_opacity_t, v_trbf)]
        with torch.cuda.device(dev):
            B.call("gs_temporal_slice_bwd", n, *[B.ptr(t) for t in ins], ctx.timestamp, *[B.ptr(v) for v in vin],
                   *[B.ptr(o) for o in outs], torch.cuda.current_stream(dev).cuda_stream)
        return (*outs, None)


def temporal_slice(
    means: Tensor,  # [N, 3]
    motion: Tensor,  # [N, 9]  linear | quadratic | cubic coefficients
    quats: Tensor,  # [N, 4]
    omega: Tensor,  # [N, 4]
    opacities: Tensor,  # [N]     activated (sigmoid applied)
    trbf_center: Tensor,  # [N, 1] or [N]
    trbf_scale: Tensor,  # [N, 1] or [N]   activated (exp applied)
    timestamp: float,
    temp_vis_mask: bool = False,
) -> Tuple[Tensor, Tensor, Tensor, Optional[Tensor]]:
    """Returns (means_t [N,3], quats_t [N,4], opacity_t [N], mask or None).

    ``mask`` (bool [N], ``trbf > 0.05``) is returned when ``temp_vis_mask`` is set; the caller filters the splats with
    it exactly as the trainer does (simple_trainer_dyngs.py:525-535)."""
    N = means.shape[0]
    assert means.shape == (N, 3) and motion.shape == (N, 9) and quats.shape == (N, 4) and omega.shape == (N, 4)
    assert opacities.shape == (N,), opacities.shape
    assert trbf_center.numel() == N and trbf_scale.numel() == N
    means_t, quats_t, opacity_t, trbf = _TemporalSlice.apply(means, motion, quats, omega, opacities, trbf_center.reshape(N),
                                                             trbf_scale.reshape(N), float(timestamp))
    mask = (trbf.detach() > TEMPORAL_VISIBILITY_THRESHOLD) if temp_vis_mask else None
    return means_t, quats_t, opacity_t, mask


# ---------------------------------------------------------------------------------------------------------------------
_RAW_BITS = {"scales": 1, "opacities": 2, "trbf_scale": 4}  # GS_DYN_RAW_* of include/gsplat_hip.h
_QUANT_SLOTS = ("scales", "quats", "opacities", "colors")   # bit k of quant_mask
_F4 = ctypes.c_float * 4


def _f32(v: float) -> float:
    return float(torch.tensor(v, dtype=torch.float32))


class DynamicSlice:
    """The temporal part of a set of dynamic gaussians, for ``rasterization(..., dynamic=DynamicSlice(...))``.

    ``rasterization(means, quats, scales, opacities, colors, ..., dynamic=ds)`` equals
    ``rasterization(*temporal_slice(means, ds.motion, quats, ds.omega, opacities, ds.trbf_center, ds.trbf_scale, ds.timestamp)...)``
    (reference examples/simple_trainer_dyngs.py:506-554) with the slice evaluated inside the projection kernels.

    raw       names among ("scales", "opacities", "trbf_scale") (or True for all three) whose tensors are the trainer's RAW
              parameters -- log-scales, opacity logits, log trbf_scale: the exp / sigmoid / exp of dyngs.py:493-505 then runs in the
              kernel too.
    quantize  {"scales" | "quats" | "opacities" | "colors": (lower, upper, bits)}: the attribute goes through the reference's
              round-to-grid STE first (gsplat/compression_simulation/ops.py:57-75): the tensor handed to ``rasterization`` is
              CLAMPED IN PLACE like ``STE.apply``'s input (hand over the parameter itself), quantized values are never
              materialised, the gradient is the identity.  (A quantized attribute that also has an activation must be "raw".)
    min_trbf  the trainer's temporal visibility filter (``temp_vis_mask``, simple_trainer_dyngs.py:526-535: splats whose temporal basis
              ``trbf`` is not above 0.05 are dropped before rasterization): such splats are CULLED by the projection (radius 0, no tiles,
              no gradient) instead of being filtered out of the arrays -- same image, and every ``meta`` tensor keeps its full [C, N]
              shape (the reference re-expands radii / depths / conics / opacities to it afterwards, :557-571).  ``t_vis_mask`` (bool [N])
              holds the mask after the call.
    A plain tuple ``(motion, omega, trbf_center, trbf_scale, timestamp)`` is accepted wherever a DynamicSlice is."""

    def __init__(self, motion: Tensor, omega: Tensor, trbf_center: Tensor, trbf_scale: Tensor, timestamp: float,
                 raw: Union[bool, Sequence[str]] = (), quantize: Optional[Dict[str, Tuple[float, float, int]]] = None,
                 min_trbf: Optional[float] = None):
        self.motion, self.omega, self.trbf_center, self.trbf_scale = motion, omega, trbf_center, trbf_scale
        self.timestamp = float(timestamp)
        self.min_trbf = None if min_trbf is None else float(min_trbf)
        self._alive: Optional[Tensor] = None
        names = tuple(_RAW_BITS) if raw is True else tuple(raw or ())
        assert all(n in _RAW_BITS for n in names), f"raw: names among {tuple(_RAW_BITS)}, got {names}"
        self.raw = names
        self.raw_mask = sum(_RAW_BITS[n] for n in set(names))
        self.quantize = dict(quantize or {})
        assert all(k in _QUANT_SLOTS for k in self.quantize), f"quantize: keys among {_QUANT_SLOTS}, got {tuple(self.quantize)}"
        for k in ("scales", "opacities"):
            assert not (k in self.quantize and k not in names), \
                f"quantize[{k!r}] works on the raw parameter (log-scales / logits): add {k!r} to raw"
        self.quant_mask = sum(1 << _QUANT_SLOTS.index(k) for k in self.quantize)
        lo, hi, rng, qn = [0.0] * 4, [0.0] * 4, [1.0] * 4, [1.0] * 4
        for k, (l_, h_, bits) in self.quantize.items():
            i = _QUANT_SLOTS.index(k)
            # python arithmetic first, then fp32, as torch does with the reference's python-float bounds (ops.py:65-70)
            lo[i], hi[i], rng[i], qn[i] = _f32(l_), _f32(h_), _f32(h_ - l_), _f32(1 / (2 ** bits - 1))
        self._tables = (_F4(*lo), _F4(*hi), _F4(*rng), _F4(*qn))

    @staticmethod
    def of(d) -> "DynamicSlice":
        return d if isinstance(d, DynamicSlice) else DynamicSlice(*d)

    def check(self, N: int) -> None:
        assert self.motion.shape == (N, 9) and self.omega.shape == (N, 4), (self.motion.shape, self.omega.shape)
        assert self.trbf_center.numel() == N and self.trbf_scale.numel() == N, (self.trbf_center.shape, self.trbf_scale.shape)

    def bind(self, quats, scales, opacities, colors, motion, omega, center, tscale):
        """Contiguous fp32 views of the four temporal tensors; refuses a quantized attribute that is not writable in place."""
        for name, t in (("quats", quats), ("scales", scales), ("opacities", opacities), ("colors", colors)):
            if name in self.quantize and t is None:
                raise RuntimeError(f"DynamicSlice: quantize[{name!r}] but no {name} reach the projection "
                                   "(more than three colour channels are quantized by the caller)")
        out = []
        for t in (motion, omega, center, tscale):
            if t.dtype != torch.float32:
                raise RuntimeError(f"DynamicSlice: expected float32 tensors, got {t.dtype}")
            out.append(t if t.is_contiguous() else t.contiguous())
        return tuple(out)

    def c_args(self, dt, fwd_on: Optional[torch.device] = None, N: int = 0):
        """The (motion ... quant_step_norm) run of arguments of gs_projection_rows_dyn_bwd; with ``fwd_on`` (a device) that of
        gs_projection_rows_dyn_fwd, which also takes the temporal visibility threshold and the mask buffer."""
        lo, hi, rng, qn = self._tables
        head = (B.ptr(dt[0]), B.ptr(dt[1]), B.ptr(dt[2]), B.ptr(dt[3]), self.timestamp)
        if fwd_on is not None:
            head += (self.min_trbf_arg(), B.ptr(self.alive_buffer(N, fwd_on)))
        return head + (self.raw_mask, self.quant_mask, ctypes.addressof(lo), ctypes.addressof(hi), ctypes.addressof(rng), ctypes.addressof(qn))

    def min_trbf_arg(self) -> float:
        return -1.0 if self.min_trbf is None else self.min_trbf

    def alive_buffer(self, N: int, device) -> Optional[Tensor]:
        """The uint8 [N] buffer the forward writes the temporal visibility into (None without ``min_trbf``)."""
        if self.min_trbf is None:
            return None
        self._alive = torch.empty((N,), dtype=torch.uint8, device=device)
        return self._alive

    @property
    def t_vis_mask(self) -> Optional[Tensor]:
        return None if self._alive is None else self._alive.bool()

    # -- the same chain through the stand-alone operators: every route the fused kernels do not cover (packed, distributed, SH colours,
    # covars, camera-pose gradients) and the reference the fused route is tested against
    def apply_unfused(self, means, quats, scales, opacities, colors):
        """-> (means_t, quats_t, scales, opacity_t, colors) for a plain ``rasterization`` call: STE hooks (in-place clamp), activations,
        ``temporal_slice``."""
        from .compression_simulation.ops import STE

        def q(name, t, act):
            if name in self.quantize:
                lo, hi, bits = self.quantize[name]
                return STE.apply(t, bits, lo, hi, act)
            if act == 1:
                return torch.exp(t)
            if act == 2:
                return torch.sigmoid(t)
            return t

        quats = q("quats", quats, 0)
        scales = q("scales", scales, 1 if "scales" in self.raw else 0)
        opacities = q("opacities", opacities, 2 if "opacities" in self.raw else 0)
        if colors is not None and "colors" in self.quantize:
            colors = q("colors", colors, 0)
        tscale = torch.exp(self.trbf_scale) if "trbf_scale" in self.raw else self.trbf_scale
        means_t, quats_t, opacity_t, trbf = _TemporalSlice.apply(means, self.motion, quats, self.omega, opacities,
                                                                 self.trbf_center.reshape(-1), tscale.reshape(-1), self.timestamp)
        if self.min_trbf is not None:
            # (the stand-alone route cannot drop the splats -- meta keeps its [C, N] shape -- so they are made invisible instead: an
            # opacity of zero composites nothing and receives no gradient, like the trainer's filter)
            self._alive = (trbf.detach() > self.min_trbf).to(torch.uint8)
            opacity_t = opacity_t * self._alive.to(opacity_t.dtype)
        return means_t, quats_t, scales, opacity_t, colors


_STG_PARTS = ("colors", "features_dir", "features_time")  # bit p of gs_stg_features_fwd's quant_mask
_F3 = ctypes.c_float * 3


class _StgFeatures(torch.autograd.Function):
    @staticmethod
    def forward(ctx, colors, features_dir, features_time, trbf_center, timestamp, quantize):
        if not colors.is_cuda:
            raise RuntimeError("stg_features: the HIP path needs device tensors (no CPU fallback)")
        n = colors.shape[0]
        for name, t in zip(_STG_PARTS, (colors, features_dir, features_time)):
            assert t.shape == (n, 3), (name, t.shape)
            if name in quantize and not (t.dtype == torch.float32 and t.is_contiguous()):
                raise RuntimeError(f"stg_features: quantize[{name!r}] clamps the tensor in place -- hand over the contiguous float32 parameter")
        parts = [t if (t.dtype == torch.float32 and t.is_contiguous()) else t.contiguous().float() for t in (colors, features_dir, features_time)]
        center = trbf_center.detach().reshape(-1).contiguous().float()
        assert center.numel() == n, trbf_center.shape
        lo, hi, rng, qn = [0.0] * 3, [0.0] * 3, [1.0] * 3, [1.0] * 3
        mask = 0
        for k, (l_, h_, bits) in quantize.items():
            i = _STG_PARTS.index(k)
            mask |= 1 << i
            lo[i], hi[i], rng[i], qn[i] = _f32(l_), _f32(h_), _f32(h_ - l_), _f32(1 / (2 ** bits - 1))
        out = torch.empty((n, 9), dtype=torch.float32, device=colors.device)
        with torch.cuda.device(colors.device):
            B.call("gs_stg_features_fwd", n, *[B.ptr(t) for t in parts], B.ptr(center), float(timestamp), mask,
                   _F3(*lo), _F3(*hi), _F3(*rng), _F3(*qn), B.ptr(out), torch.cuda.current_stream(colors.device).cuda_stream)
        ctx.save_for_backward(center)
        ctx.timestamp = float(timestamp)
        return out

    @staticmethod
    def backward(ctx, v_out):
        (center,) = ctx.saved_tensors
        n = center.shape[0]
        need = ctx.needs_input_grad
        v_out = v_out.contiguous().float()
        grads = [torch.empty((n, 3), dtype=torch.float32, device=center.device) if need[i] else None for i in range(3)]
        with torch.cuda.device(center.device):
            B.call("gs_stg_features_bwd", n, B.ptr(v_out), B.ptr(center), ctx.timestamp, *[B.ptr(g) for g in grads],
                   torch.cuda.current_stream(center.device).cuda_stream)
        return (*grads, None, None, None)


def stg_features(colors: Tensor, features_dir: Tensor, features_time: Tensor, trbf_center: Tensor, timestamp: float,
                 quantize: Optional[Dict[str, Tuple[float, float, int]]] = None) -> Tensor:
    """The spacetime trainer's nine colour channels ``torch.cat((colors, features_dir, tforpoly * features_time), dim=1)`` with
    ``tforpoly = (timestamp - trbf_center).detach()`` (reference examples/simple_trainer_STG.py:506-551) in one kernel each way
    (``gs_stg_features_fwd`` / ``_bwd``, csrc/dynamic.hip) -- same values bit for bit, no [N,9] copy by ``cat``, no split + copies by
    autograd on the way back.  ``quantize``: {"colors" | "features_dir" | "features_time": (lower, upper, bits)} runs that part through
    the compression simulation's round-to-grid STE first (the tensor is CLAMPED IN PLACE like ``STE.apply``'s input; identity gradient)."""
    quantize = dict(quantize or {})
    assert all(k in _STG_PARTS for k in quantize), f"quantize: keys among {_STG_PARTS}, got {tuple(quantize)}"
    return _StgFeatures.apply(colors, features_dir, features_time, trbf_center, timestamp, quantize)


def render_dynamic(splats: Dict[str, Tensor], timestamp: float, viewmats: Tensor, Ks: Tensor, width: int, height: int,
                   compression_sim=None, step: int = 0, features: str = "colors", temp_vis_mask: bool = False, **kwargs):
    """The dynamic trainer's ``rasterize_splats`` (reference examples/simple_trainer_dyngs.py:463-577, compression_sim on or off) on
    the fused route: ``splats`` is the trainer's RAW parameter dict (means, scales (log), quats, opacities (logits), trbf_center,
    trbf_scale (log), motion, omega, colors [, features_dir, features_time]); returns ``(render_colors, render_alphas, info)``.

    With ``compression_sim`` (an ``STGCompressionSimulation`` in "round" mode) the hooks of scales / quats / opacities / colors run
    inside the projection kernel (parameters clamped in place, as the hooks do) unless the step needs their quantized values for the
    bits estimator (``step > entropy_steps[name]``) -- those attributes, and every other mode of the simulation, go through
    ``simulate_compression`` as usual and only the activations + slice are fused.  ``features="stg"`` renders the spacetime trainer's
    nine channels cat(colors, features_dir, tau * features_time) (examples/simple_trainer_STG.py:506-551).
    ``temp_vis_mask`` (the trainer's option, dyngs.py:137, 526-571): splats whose temporal basis is <= 0.05 at this timestamp are culled
    by the projection; ``info["t_vis_mask"]`` is the mask and the per-gaussian ``info`` tensors are full-size, as the reference leaves them
    (its ``info["means2d"]`` alone stays compacted to the masked subset: here it is [C, N, 2] like the rest).
    -> also returns ``esti_bits`` of the hooks that ran outside as ``info["esti_bits"]``."""
    from .rendering import rasterization

    P = dict(splats)
    esti_bits = {}
    quantize = {}
    raw = ["scales", "opacities", "trbf_scale"]
    sim = compression_sim
    if sim is not None:
        in_kernel = []
        if getattr(sim, "q_type", None) == "round":
            # (the nine-channel render takes its three colour parts through stg_features: their hooks ride in THAT kernel)
            for name in _QUANT_SLOTS + (("features_dir", "features_time") if (features == "stg" and P["colors"].is_cuda) else ()):
                if name == "colors" and features != "colors" and not P["colors"].is_cuda:
                    continue
                needs_bits = (sim.entropy_model_enable and sim.entropy_model_option.get(name, False)
                              and step > sim.entropy_steps.get(name, -1) and sim.entropy_models.get(name) is not None)
                if sim.simulation_option.get(name, False) and sim.bds.get(name) is not None and not needs_bits and name in P:
                    in_kernel.append(name)
        rest = {k: v for k, v in P.items() if k not in in_kernel and sim.simulation_option.get(k, False)}
        if rest:
            new, bits = sim.simulate_compression(rest, step)
            P.update(new)
            esti_bits.update(bits)
        for name in in_kernel:
            lo, hi = sim.bds[name]
            quantize[name] = (lo, hi, sim.q_bitwidth[name])
    if features == "stg":
        if P["colors"].is_cuda:
            colors = stg_features(P["colors"], P["features_dir"], P["features_time"], P["trbf_center"], timestamp,
                                  quantize={k: quantize.pop(k) for k in _STG_PARTS if k in quantize})
        else:
            tau = (float(timestamp) - P["trbf_center"]).detach()
            colors = torch.cat((P["colors"], P["features_dir"], tau * P["features_time"]), dim=1)
    else:
        colors = P["colors"]
    ds = DynamicSlice(P["motion"], P["omega"], P["trbf_center"], P["trbf_scale"], timestamp, raw=raw, quantize=quantize,
                      min_trbf=TEMPORAL_VISIBILITY_THRESHOLD if temp_vis_mask else None)
    rc, ra, info = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], colors, viewmats, Ks, width, height,
                                 dynamic=ds, **kwargs)
    info["esti_bits"] = esti_bits
    if temp_vis_mask:
        info["t_vis_mask"] = ds.t_vis_mask
    return rc, ra, info
