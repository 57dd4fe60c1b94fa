"""What the ORDER of the splats in memory is worth at BASELINE config 2 (1,006,065 gaussians, SH degree 3, one 1080p camera, forward +
backward): the fixture's own order (the garden crop tiled scene_grid x scene_grid), a random shuffle (the worst case: what
bench.py --dynamic uses by default), and the same splats along a Z-order curve of their means (compression.morton_order -- the
order the codec's grid sort leaves behind, and one a trainer can restore at densification time).  Results are the same up to the
summation order of the atomics; only the memory traffic of the per-splat stages changes.
usage: python tools/bench_order.py [steps]"""
import gc
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gscodec_studio_amd import rasterization  # noqa: E402
from gscodec_studio_amd._helper import sh_workload  # noqa: E402
from gscodec_studio_amd.compression import morton_order  # noqa: E402

gc.collect()
gc.freeze()


def run(order, steps):
    w = sh_workload(scene_grid=3, device="cuda")
    N = w["N"]
    if order == "shuffle":
        perm = torch.randperm(N, generator=torch.Generator().manual_seed(7)).cuda()
    elif order == "morton":
        perm = morton_order(w["means"])
    else:
        perm = None
    keys = ("means", "quats", "scales", "opacities", "sh")
    P = [(w[k] if perm is None else w[k][perm]).contiguous().clone().requires_grad_(True) for k in keys]

    def step():
        for p in P:
            p.grad = None
        rc, ra, meta = rasterization(*P, w["viewmats"], w["Ks"], w["width"], w["height"], sh_degree=3, packed=False)
        rc.sum().backward()
        return meta

    for _ in range(10):
        meta = step()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    print(f"order {order:8s} N = {N}  I = {meta['flatten_ids'].numel()}  {best:.4f} ms/step = {N / best / 1e3:7.1f} Msplats/s", flush=True)


if __name__ == "__main__":
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    for order in ("fixture", "shuffle", "morton", "fixture"):
        run(order, steps)
