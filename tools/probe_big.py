"""Sizes beyond the BASELINE configs (scene_grid 9 / 13 / 17: 9 M / 18.9 M / 32 M gaussians, one 1080p camera): the fast path runs, and
agrees with the packed route (other projection / binning kernels, same compositing) on image and gradients."""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gscodec_studio_amd import rasterization  # noqa: E402
from gscodec_studio_amd._helper import sh_workload  # noqa: E402

gc.collect()
gc.freeze()
for grid in [int(a) for a in sys.argv[1:]] or [9, 13]:
    w = sh_workload(scene_grid=grid, device="cuda")
    N = w["N"]
    outs = []
    for packed in (False, True):
        P = [w[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")]
        for _ in range(3):
            for p in P:
                p.grad = None
            rc, ra, meta = rasterization(*P, w["viewmats"], w["Ks"], w["width"], w["height"], sh_degree=3, packed=packed)
            rc.sum().backward()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            for p in P:
                p.grad = None
            rc, ra, meta = rasterization(*P, w["viewmats"], w["Ks"], w["width"], w["height"], sh_degree=3, packed=packed)
            rc.sum().backward()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 10 * 1e3
        outs.append((rc.detach(), ra.detach(), [p.grad.clone() for p in P]))
        print(f"grid {grid} N = {N:,} packed={packed}: I = {meta['flatten_ids'].numel():,}  {ms:.3f} ms/step = {N / ms / 1e3:.0f} Msplats/s  "
              f"peak {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB", flush=True)
    (rc0, ra0, g0), (rc1, ra1, g1) = outs
    d_img = float((rc0 - rc1).abs().max())
    rel = [float((a - b).norm() / b.norm().clamp_min(1e-30)) for a, b in zip(g0, g1)]
    print(f"   unpacked vs packed: max |d image| {d_img:.2e}, gradient rel L2 {['%.1e' % r for r in rel]}", flush=True)
    assert d_img < 1e-4 and max(rel) < 1e-3
    del outs, w
    torch.cuda.empty_cache()
