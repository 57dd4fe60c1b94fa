"""Compositing kernels on a synthetic UNIFORM scene (every tile has about the same list length) with the same pair count
as bench config 2 -- separates load imbalance / tail effects from steady-state throughput.
usage: python tools/bench_uniform.py [opacity_hi]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gscodec_studio_amd import isect_offset_encode, isect_tiles, rasterize_to_pixels  # noqa: E402

import gc  # noqa: E402

# a full collection over the ~10^5 objects torch's import leaves behind takes 30-50 ms and lands in the middle of a timed loop
# (one 33 ms call in 30: a "2.1 ms" forward that is 0.44): park them in the permanent generation
gc.collect()
gc.freeze()

dev = torch.device("cuda")
W, H, ts = 1920, 1080, 16
tw, th = W // ts, (H + ts - 1) // ts
g = torch.Generator(device=dev).manual_seed(0)
op_hi = float(sys.argv[1]) if len(sys.argv) > 1 else 0.9
n = 292931
means2d = torch.rand(1, n, 2, device=dev, generator=g) * torch.tensor([W, H], device=dev)
sig = 3.0 + 6.0 * torch.rand(1, n, device=dev, generator=g)  # px
radii = torch.ceil(3.0 * sig).to(torch.int32)
conics = torch.stack([1 / sig**2, torch.zeros_like(sig), 1 / sig**2], -1)
depths = torch.rand(1, n, device=dev, generator=g) * 10 + 0.5
opac = (0.05 + (op_hi - 0.05) * torch.rand(1, n, device=dev, generator=g))
colors = torch.rand(1, n, 3, device=dev, generator=g)
tpg, ids, flat = isect_tiles(means2d, radii, depths, ts, tw, th)
offs = isect_offset_encode(ids, 1, tw, th)
cnt = torch.diff(torch.cat([offs.flatten(), torch.tensor([ids.numel()], device=dev, dtype=torch.int32)]))
print(f"I = {ids.numel()}, per tile mean {cnt.float().mean():.0f} max {int(cnt.max())}")
m2 = means2d.clone().requires_grad_(True)
cn = conics.clone().requires_grad_(True)
co = colors.clone().requires_grad_(True)
op = opac.clone().requires_grad_(True)


def run():
    rc, ra = rasterize_to_pixels(m2, cn, co, op, W, H, ts, offs, flat)
    return rc, ra


for _ in range(3):
    rc, ra = run()
    rc.sum().backward()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
tf = tb = 0.0
for _ in range(10):
    ev[0].record()
    rc, ra = run()
    ev[1].record()
    rc.sum().backward()
    ev[2].record()
    torch.cuda.synchronize()
    tf += ev[0].elapsed_time(ev[1])
    tb += ev[1].elapsed_time(ev[2])
print(f"alpha mean {float(ra.mean()):.3f}; fwd {tf / 10:.3f} ms  bwd(+loss glue) {tb / 10:.3f} ms")
