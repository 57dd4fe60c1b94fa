"""CPU oracle for the unfused public ops -- TEST INFRASTRUCTURE ONLY (tests/ and smoke() may import it).

Restates, in float64 torch on the CPU (gradients by autograd on this restatement),
  * world_to_cam   -- gsplat/cuda/include/transform.cuh:8-46 (p_c = R p + t, S_c = R S R^T),
  * proj           -- gsplat/cuda/include/proj.cuh: ortho 9-37, pinhole 80-119 (the x/z, y/z clamp to
                      +-(lim + 0.3 tan_fov) enters the Jacobian only), fisheye 202-243,
  * rasterize_to_indices_in_range -- gsplat/cuda/csrc/rasterize_to_indices_in_range.cu:16-175
                      (batches of tile_size^2 sorted entries, alpha = min(0.999, o exp(-sigma)), skip
                      sigma < 0 or alpha < 1/255, exclusive stop at T (1 - alpha) <= 1e-4), as plain
                      Python loops for small cases.
Pinned by tests/golden/make_golden_unfused.py against the reference's _torch_impl
(_world_to_cam, _persp_proj, _ortho_proj, _fisheye_proj) outputs and autograd gradients.  The
indices op has no CPU-runnable reference (CUDA only): it is "parity unpinned" by reference code and
is cross-checked against the compositing forward instead (tests/test_gpu_unfused.py).
"""
from __future__ import annotations

import numpy as np
import torch


def world_to_cam(means, covars, viewmats):
    R = viewmats[:, :3, :3]
    t = viewmats[:, :3, 3]
    means_c = (R[:, None] @ means[None, :, :, None])[..., 0] + t[:, None]
    covars_c = R[:, None] @ covars[None] @ R[:, None].transpose(-1, -2)
    return means_c, covars_c


def _jacobian(means, Ks, width, height, model):
    x, y, z = means.unbind(-1)
    fx, fy = Ks[:, 0, 0][:, None], Ks[:, 1, 1][:, None]
    cx, cy = Ks[:, 0, 2][:, None], Ks[:, 1, 2][:, None]
    zero = torch.zeros_like(x)
    if model == "ortho":
        J = torch.stack([fx + zero, zero, zero, zero, fy + zero, zero], -1)
        m = torch.stack([fx * x + cx, fy * y + cy], -1)
    elif model == "pinhole":
        tan_x, tan_y = 0.5 * width / fx, 0.5 * height / fy
        lim_xp, lim_xn = (width - cx) / fx + 0.3 * tan_x, cx / fx + 0.3 * tan_x
        lim_yp, lim_yn = (height - cy) / fy + 0.3 * tan_y, cy / fy + 0.3 * tan_y
        rz = 1.0 / z
        txc = z * torch.minimum(lim_xp, torch.maximum(-lim_xn, x * rz))
        tyc = z * torch.minimum(lim_yp, torch.maximum(-lim_yn, y * rz))
        J = torch.stack([fx * rz, zero, -fx * txc * rz * rz, zero, fy * rz, -fy * tyc * rz * rz], -1)
        m = torch.stack([fx * x * rz + cx, fy * y * rz + cy], -1)
    elif model == "fisheye":
        eps = 0.0000001
        x2, y2, xy = x * x + eps, y * y, x * y
        r2 = x2 + y2
        inv_rho = 1.0 / (r2 + z * z)
        ln = torch.sqrt(x * x + y * y) + eps
        b = torch.atan2(ln, z) / ln / r2
        a = z * inv_rho / r2
        J = torch.stack([fx * (x2 * a + y2 * b), fx * xy * (a - b), -fx * x * inv_rho,
                         fy * xy * (a - b), fy * (y2 * a + x2 * b), -fy * y * inv_rho], -1)
        th = torch.atan2(ln, z + eps)
        m = torch.stack([x * fx * th / ln + cx, y * fy * th / ln + cy], -1)
    else:
        raise ValueError(model)
    return J.reshape(*x.shape, 2, 3), m


def proj(means, covars, Ks, width, height, model="pinhole"):
    J, m = _jacobian(means, Ks, width, height, model)
    return m, J @ covars @ J.transpose(-1, -2)


def with_grads(fn, inputs, v_outs):
    """Run ``fn`` in float64 with autograd; returns (outputs, grads) as numpy arrays."""
    ins = [torch.tensor(np.asarray(a), dtype=torch.float64, requires_grad=True) for a in inputs]
    outs = fn(*ins)
    loss = sum((o * torch.tensor(np.asarray(v), dtype=torch.float64)).sum() for o, v in zip(outs, v_outs))
    grads = torch.autograd.grad(loss, ins, allow_unused=True)
    return [o.detach().numpy() for o in outs], [None if g is None else g.numpy() for g in grads]


def rasterize_to_indices_in_range(range_start, range_end, transmittances, means2d, conics, opacities, width, height,
                                  tile_size, isect_offsets, flatten_ids):
    """Plain loops (float32 arithmetic like the kernel).  Returns (gaussian_ids, pixel_ids, camera_ids) int64."""
    f = np.float32
    C, N = means2d.shape[:2]
    th, tw = isect_offsets.shape[1:]
    offs = np.concatenate([isect_offsets.reshape(-1), [len(flatten_ids)]]).astype(np.int64)
    m2 = means2d.reshape(-1, 2).astype(f)
    cn = conics.reshape(-1, 3).astype(f)
    op = opacities.reshape(-1).astype(f)
    B = tile_size * tile_size
    g_out, p_out, c_out = [], [], []
    for c in range(C):
        for i in range(height):
            for j in range(width):
                lin = (c * th + i // tile_size) * tw + j // tile_size
                rs, re = offs[lin], offs[lin + 1]
                nb = (re - rs + B - 1) // B
                if range_start >= nb:
                    continue
                lo = rs + B * range_start
                hi = min(re, rs + B * min(range_end, nb))
                T = f(transmittances[c, i, j])
                px, py = f(j + 0.5), f(i + 0.5)
                for idx in range(lo, hi):
                    g = int(flatten_ids[idx])
                    dx, dy = m2[g, 0] - px, m2[g, 1] - py
                    sigma = f(0.5) * (cn[g, 0] * dx * dx + cn[g, 2] * dy * dy) + cn[g, 1] * dx * dy
                    alpha = min(f(0.999), op[g] * np.exp(-sigma, dtype=f))
                    if sigma < 0 or alpha < f(1.0 / 255.0):
                        continue
                    nT = T * (f(1.0) - alpha)
                    if nT <= f(1e-4):
                        break
                    g_out.append(g % N)
                    p_out.append(i * width + j)
                    c_out.append(c)
                    T = nT
    return np.array(g_out, np.int64), np.array(p_out, np.int64), np.array(c_out, np.int64)


def temporal_slice(means, motion, quats, omega, opacities, trbf_center, trbf_scale, timestamp):
    """examples/simple_trainer_dyngs.py:506-521 restated (float64 torch; ``tau`` detached where the trainer detaches
    ``tforpoly``).  Pinned by tests/golden/make_golden_dynamic.py: the trainer script cannot be imported here
    (tyro / nerfview / datasets missing), so that script extracts the method's statements with ``ast``, executes them on CPU
    tensors and asserts this function reproduces outputs and gradients (1e-12 in float64); tests/golden/dynamic.npz."""
    tau = timestamp - trbf_center.reshape(-1)
    trbf = torch.exp(-((tau / (2.0 ** 0.5 * trbf_scale.reshape(-1))) ** 2))
    opacity = opacities * trbf
    tp = tau.detach()[:, None]
    means_t = means + motion[:, 0:3] * tp + motion[:, 3:6] * tp * tp + motion[:, 6:9] * tp * tp * tp
    x = quats + tp * omega
    quats_t = x / torch.clamp(x.norm(dim=-1, keepdim=True), min=1e-12)
    return means_t, quats_t, opacity, trbf


def accumulate(means2d, conics, opacities, colors, gaussian_ids, pixel_ids, camera_ids, image_width, image_height):
    """gsplat/cuda/_torch_impl.py:432-519 restated in torch (any dtype; differentiable): the reference's own statements for the alphas
    (490-501) and its two nerfacc calls (503-517) from nerfacc's published definitions -- ``render_weight_from_alpha``: weight = alpha x
    the exclusive product of (1 - alpha) along each ray; ``accumulate_along_rays``: ``index_add`` of weight (x value) into the rays.
    nerfacc is absent here and unpinned by the reference (examples/requirements.txt:9-10: the git head), so this restatement is PARITY
    UNPINNED by executable reference code; tests pin it against a per-ray python loop (tests/test_unfused_cpu.py) and, composed with
    rasterize_to_indices_in_range as in ``_rasterize_to_pixels`` (522-617), against the compositing oracle.
    A ray = a maximal run of consecutive entries with equal (camera, pixel), as nerfacc's packed scans see sorted ray_indices."""
    C, N = means2d.shape[:2]
    channels = colors.shape[-1]
    g, p, c = (torch.as_tensor(np.asarray(t), dtype=torch.int64) for t in (gaussian_ids, pixel_ids, camera_ids))
    px = (p % image_width).to(means2d.dtype) + 0.5
    py = (p // image_width).to(means2d.dtype) + 0.5
    deltas = torch.stack([px, py], -1) - means2d[c, g]
    cn = conics[c, g]
    sigmas = 0.5 * (cn[:, 0] * deltas[:, 0] ** 2 + cn[:, 2] * deltas[:, 1] ** 2) + cn[:, 1] * deltas[:, 0] * deltas[:, 1]
    alphas = torch.clamp_max(opacities[c, g] * torch.exp(-sigmas), 0.999)
    rays = c * image_height * image_width + p
    M = rays.shape[0]
    total = C * image_height * image_width
    if M == 0:
        return means2d.new_zeros((C, image_height, image_width, channels)), means2d.new_zeros((C, image_height, image_width, 1))
    head = torch.ones(M, dtype=torch.bool)
    head[1:] = rays[1:] != rays[:-1]
    run = torch.cumsum(head.to(torch.int64), 0) - 1                      # run index of every entry
    start = torch.nonzero(head).reshape(-1)
    pos = torch.arange(M) - start[run]                                     # position inside the run
    L = int(pos.max()) + 1
    one_minus = alphas.new_ones((start.numel(), L + 1))
    one_minus = one_minus.index_put((run, pos + 1), 1.0 - alphas)          # column 0 = 1: exclusive product
    trans = torch.cumprod(one_minus, dim=1)[run, pos]
    weights = alphas * trans
    renders = means2d.new_zeros((total, channels)).index_add(0, rays, weights[:, None] * colors[c, g])
    acc = means2d.new_zeros((total,)).index_add(0, rays, weights)
    return renders.reshape(C, image_height, image_width, channels), acc.reshape(C, image_height, image_width, 1)
