"""Render modes and options around the headline call at BASELINE config 2 (1 M gaussians, SH 3, one 1080p camera, forward + backward):
what each costs next to plain RGB.  usage: python tools/bench_modes.py"""
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
w = sh_workload(scene_grid=3, device="cuda")
P = [w[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")]
bg = torch.rand(1, 3, device="cuda")


def run(tag, steps=100, **kw):
    def step():
        for p in P:
            p.grad = None
        rc, ra, meta = rasterization(*P, w["viewmats"], w["Ks"], w["width"], w["height"], sh_degree=3, **kw)
        rc.sum().backward()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    print(f"{tag:44s} {best:.4f} ms/step", flush=True)


run("RGB (headline call)", packed=False)
run("RGB + backgrounds", packed=False, backgrounds=bg)
run("RGB+ED (depth loss)", packed=False, render_mode="RGB+ED")
run("RGB+D", packed=False, render_mode="RGB+D")
run("antialiased", packed=False, rasterize_mode="antialiased")
run("absgrad", packed=False, absgrad=True)
run("packed=True", packed=True)
run("packed=True, sparse_grad", packed=True, sparse_grad=True)
run("tile_size=8", packed=False, tile_size=8)
run("RGB (again)", packed=False)
