"""Dynamic (spacetime) gaussians: temporal slicing at one timestamp, fused.

``temporal_slice`` is the elementwise chain that the reference's dynamic-scene trainer runs in front of
``rasterization()`` (examples/simple_trainer_dyngs.py:506-536; the viewer's copy:
examples/simple_viewer_dyn.py:84-101), as ONE HIP kernel each way (csrc/dynamic.hip) instead of ~25 torch
kernels.  Same inputs as the trainer's local variables: ``opacities`` and ``trbf_scale`` are the
ACTIVATED values (sigmoid / exp are applied by the caller, as in the trainer).  No torch fallback.
"""
from __future__ import annotations

from typing import Optional, Tuple

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
