"""Synthetic workload generator: counterpart of the reference's ``gsplat/_helper.py:9-55``
(``load_test_data``) with the randomness made reproducible.

``load_test_data(scene_grid=g)`` tiles the garden crop (111,785 points, shipped as
``assets/garden_crop.npz``, derived data of the reference's assets/test_garden.npz) on a g x g
grid => N = g^2 * 111,785 gaussians, and draws scales ~ U(0, 0.02)^3, quats = normalize(N(0,1)^4),
opacities ~ U(0,1).  The reference draws them from the global RNG unseeded; here a CPU
generator with an explicit seed is used, so CPU (oracle) and GPU runs see identical inputs.
``sh_workload`` adds degree-3 SH coefficients (BASELINE config 2).
"""
from __future__ import annotations

import os
from typing import Dict, Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

_ASSET = os.path.join(os.path.dirname(os.path.abspath(__file__)), "assets", "garden_crop.npz")
SH_C0 = 0.2820947917738781


def load_test_data(
    data_path: Optional[str] = None,
    device="cuda",
    scene_crop: Tuple[float, float, float, float, float, float] = (-2, -2, -2, 2, 2, 2),
    scene_grid: int = 1,
    seed: int = 42,
):
    """Returns (means, quats, scales, opacities, colors, viewmats, Ks, width, height)."""
    assert scene_grid % 2 == 1, "scene_grid must be odd"
    data = np.load(data_path or _ASSET)
    height, width = int(data["height"]), int(data["width"])
    viewmats = torch.from_numpy(data["viewmats"]).float()
    Ks = torch.from_numpy(data["Ks"]).float()
    means = torch.from_numpy(data["means3d"]).float()
    colors = torch.from_numpy(data["colors"] / 255.0).float()

    aabb = torch.tensor(scene_crop, dtype=torch.float32)
    edges = aabb[3:] - aabb[:3]
    sel = ((means >= aabb[:3]) & (means <= aabb[3:])).all(dim=-1)
    means, colors = means[sel], colors[sel]

    r = scene_grid
    gx, gy = torch.meshgrid(torch.arange(-(r // 2), r // 2 + 1), torch.arange(-(r // 2), r // 2 + 1), indexing="ij")
    grid = torch.stack([gx, gy, torch.zeros_like(gx)], dim=-1).reshape(-1, 3).float()
    means = (means[None, :, :] + grid[:, None, :] * edges[None, None, :]).reshape(-1, 3)
    colors = colors.repeat(r**2, 1)

    N = len(means)
    g = torch.Generator().manual_seed(seed)
    scales = torch.rand((N, 3), generator=g) * 0.02
    quats = F.normalize(torch.randn((N, 4), generator=g), dim=-1)
    opacities = torch.rand((N,), generator=g)
    out = (means, quats, scales, opacities, colors, viewmats, Ks)
    out = tuple(t.contiguous().to(device) for t in out)
    return out + (width, height)


def rescale_intrinsics(Ks: torch.Tensor, width: int, height: int, new_width: int, new_height: int) -> torch.Tensor:
    """profiling/main.py:85-87 of the reference: scale K rows to the new resolution."""
    Ks = Ks.clone()
    Ks[..., 0, :] *= new_width / width
    Ks[..., 1, :] *= new_height / height
    return Ks


def sh_workload(scene_grid: int = 3, width: int = 1920, height: int = 1080, n_cameras: int = 1, sh_degree: int = 3,
                device="cuda", seed: int = 42, camera_mode: str = "fixture") -> Dict:
    """BASELINE.json config 2: scene_grid=3 -> N = 1,006,065 gaussians, SH degree 3, 1080p.

    Cameras, ``camera_mode="fixture"``: the fixture's 3 cameras, cycled when n_cameras > 3 with a small yaw so
    that every camera of a batch is distinct.  ``"jitter0"``: camera 0 (the one config 2 is quoted on) for
    every slot, yawed by +-0.01 rad * slot -- equal work per camera, which is what a weak-scaling run over
    camera-sharded ranks needs (the three fixture cameras differ ~2x in visible splats).
    """
    means, quats, scales, opacities, rgb, viewmats, Ks, w0, h0 = load_test_data(device="cpu", scene_grid=scene_grid, seed=seed)
    Ks = rescale_intrinsics(Ks, w0, h0, width, height)
    K = (sh_degree + 1) ** 2
    N = means.shape[0]
    g = torch.Generator().manual_seed(seed + 1)
    sh = torch.empty((N, K, 3))
    sh[:, 0] = (rgb - 0.5) / SH_C0
    if K > 1:
        sh[:, 1:] = torch.randn((N, K - 1, 3), generator=g) * 0.05
    vm, kk = [], []
    for i in range(n_cameras):
        jitter = camera_mode == "jitter0"
        V = viewmats[0 if jitter else i % 3].clone()
        if jitter or i >= 3:
            a = (0.01 * i if jitter else 0.05 * (i // 3)) * (1 if i % 2 else -1)
            c, s = float(np.cos(a)), float(np.sin(a))
            Rz = torch.tensor([[c, -s, 0, 0], [s, c, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]], dtype=torch.float32)
            V = V @ Rz
        vm.append(V)
        kk.append(Ks[0 if jitter else i % 3])
    d = dict(means=means, quats=quats, scales=scales, opacities=opacities, sh=sh, rgb=rgb,
             viewmats=torch.stack(vm), Ks=torch.stack(kk))
    d = {k: v.contiguous().to(device) for k, v in d.items()}
    d.update(width=width, height=height, sh_degree=sh_degree, N=N)
    return d


DYNAMIC_KEYS = ("means", "scales", "quats", "opacities", "trbf_center", "trbf_scale", "motion", "omega", "colors", "features_dir",
                "features_time")


def dynamic_workload(n_splats: int = 2_000_000, width: int = 1920, height: int = 1080, device="cuda", seed: int = 42,
                     order: str = "shuffle") -> Dict:
    """BASELINE.json config 5, one frame of it: ``n_splats`` DYNAMIC (spacetime) gaussians with the trainer's raw parameter set
    (reference examples/simple_trainer_dyngs.py:283-332: means, log-scales, quats, opacity logits, trbf_center, log trbf_scale,
    motion [N,9], omega [N,4], colors / features_dir / features_time [N,3]) and one 1080p camera.  The static part is the
    ``load_test_data`` scene (the smallest odd grid with at least ``n_splats`` gaussians, the first ``n_splats`` of a seeded
    shuffle); the temporal part is drawn so that, as in a trained spacetime scene, only part of the splats is alive at a given
    timestamp: centres ~ U(0,1), log-scale ~ U(-1.5, 0.5), motion ~ 0.02 N(0,1), omega ~ 0.1 N(0,1).
    ``order``: "shuffle" (default: the arbitrary order a trained scene's splats are in) or "morton" (the same splats sorted along a
    Z-order curve of their positions: neighbours in space are neighbours in memory -- what a trainer could do at densification time)."""
    grid = 1
    while grid * grid * 111_785 < n_splats:
        grid += 2
    means, quats, scales, opacities, rgb, viewmats, Ks, w0, h0 = load_test_data(device="cpu", scene_grid=grid, seed=seed)
    g = torch.Generator().manual_seed(seed + 2)
    sel = torch.randperm(means.shape[0], generator=g)[:n_splats]
    n = int(sel.numel())
    Ks = rescale_intrinsics(Ks, w0, h0, width, height)
    op = opacities[sel].clamp(1e-4, 1 - 1e-4)
    d = dict(
        means=means[sel], scales=scales[sel].clamp_min(1e-6).log(), quats=quats[sel], opacities=torch.log(op / (1 - op)),
        trbf_center=torch.rand((n, 1), generator=g), trbf_scale=torch.rand((n, 1), generator=g) * 2.0 - 1.5,
        motion=0.02 * torch.randn((n, 9), generator=g), omega=0.1 * torch.randn((n, 4), generator=g),
        colors=rgb[sel], features_dir=torch.randn((n, 3), generator=g), features_time=torch.randn((n, 3), generator=g),
        viewmats=viewmats[:1], Ks=Ks[:1])
    if order == "morton":
        m = d["means"]
        q = ((m - m.min(0).values) / (m.max(0).values - m.min(0).values).clamp_min(1e-9) * 1023).long().clamp(0, 1023)
        code = torch.zeros(n, dtype=torch.long)
        for b in range(10):
            for a in range(3):
                code |= ((q[:, a] >> b) & 1) << (3 * b + a)
        perm = torch.argsort(code)
        for k in DYNAMIC_KEYS:
            d[k] = d[k][perm]
    else:
        assert order == "shuffle", order
    d = {k: v.contiguous().float().to(device) for k, v in d.items()}
    d.update(width=width, height=height, N=n, scene_grid=grid, order=order)
    return d
