import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.getcwd()))
from gscodec_studio_amd import rasterization
from gscodec_studio_amd._helper import sh_workload
w = sh_workload(scene_grid=17, device="cuda")
P = [w[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")]
for _ in range(8):
    for p in P: p.grad = None
    rc, ra, meta = rasterization(*P, w["viewmats"], w["Ks"], w["width"], w["height"], sh_degree=3, packed=False)
    rc.sum().backward()
torch.cuda.synchronize()
