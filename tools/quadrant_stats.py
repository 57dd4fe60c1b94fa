"""How many 8x8 quadrants / 16x8 half tiles of its tile a (tile, splat) pair touches on the bench scene (bounding box of the 3-sigma
radius against the quadrant rectangles: an upper bound of what the kernels' exact ellipse test keeps).  Input to the costing of a
one-pass 17..32-channel backward with two waves per tile (profiles/round6_notes.md section 4)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gscodec_studio_amd import rasterization  # noqa: E402
from gscodec_studio_amd._helper import sh_workload  # noqa: E402

w = sh_workload(scene_grid=3, device="cuda")
with torch.no_grad():
    _, _, meta = rasterization(w["means"], w["quats"], w["scales"], w["opacities"], w["sh"], w["viewmats"], w["Ks"], 1920, 1080, sh_degree=3, packed=False)
ids, flat = meta["isect_ids"], meta["flatten_ids"].long()
tile = (ids >> 32) & ((1 << 14) - 1)  # one camera: tile id in the bits above the depth
tw = meta["tile_width"]
tx, ty = (tile % tw).float() * 16, (tile // tw).float() * 16
m = meta["means2d"].reshape(-1, 2)[flat]
r = meta["radii"].reshape(-1)[flat].float()
x0, x1, y0, y1 = m[:, 0] - r, m[:, 0] + r, m[:, 1] - r, m[:, 1] + r
def hit(ax0, ax1, ay0, ay1):
    return (x1 > ax0) & (x0 < ax1) & (y1 > ay0) & (y0 < ay1)
q = torch.stack([hit(tx + 8 * (i & 1), tx + 8 * (i & 1) + 8, ty + 8 * (i >> 1), ty + 8 * (i >> 1) + 8) for i in range(4)], 1)
nq = q.sum(1).float()
top, bot = q[:, 0] | q[:, 1], q[:, 2] | q[:, 3]
nh = top.float() + bot.float()
print(f"pairs {ids.numel()}: quadrants touched per pair {nq.mean():.2f} (1: {(nq == 1).float().mean():.2f}, 2: {(nq == 2).float().mean():.2f}, "
      f"3: {(nq == 3).float().mean():.2f}, 4: {(nq == 4).float().mean():.2f}); 16x8 halves touched per pair {nh.mean():.2f}")
