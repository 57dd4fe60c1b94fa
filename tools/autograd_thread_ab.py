import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gscodec_studio_amd import rasterization
from gscodec_studio_amd._helper import sh_workload
w = sh_workload(scene_grid=3, device="cuda:0")
P = {k: w[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")}
vm, Ks = w["viewmats"][:1].contiguous(), w["Ks"][:1].contiguous()
mt = os.environ.get("MT", "1") == "1"
def step():
    for p in P.values(): p.grad = None
    rc, ra, meta = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["sh"], vm, Ks, 1920, 1080, sh_degree=3, packed=False)
    with torch.autograd.set_multithreading_enabled(mt):
        rc.sum().backward()
for _ in range(30): step()
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(50): step()
    torch.cuda.synchronize(); print(f"multithreading={mt}: {(time.perf_counter()-t0)/50*1e3:.4f} ms/step")
