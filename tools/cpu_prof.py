import os, sys, time, cProfile, pstats, io
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from gscodec_studio_amd import rasterization
from gscodec_studio_amd._helper import sh_workload
dev = torch.device("cuda:0")
w = sh_workload(scene_grid=3, device=dev)
params = {k: w[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")}
vm, Ks = w["viewmats"][:1].contiguous(), w["Ks"][:1].contiguous()
def step():
    for p in params.values(): p.grad = None
    rc, ra, meta = rasterization(params["means"], params["quats"], params["scales"], params["opacities"], params["sh"], vm, Ks, 1920, 1080, sh_degree=3, packed=False)
    rc.sum().backward()
for _ in range(5): step()
torch.cuda.synchronize()
K = 30
t0 = time.perf_counter()
for _ in range(K): step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"cpu-side per step {1e3*(t1-t0)/K:.3f} ms ; wall per step {1e3*(t2-t0)/K:.3f} ms")
pr = cProfile.Profile(); pr.enable()
for _ in range(K): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(18); print(s.getvalue()[:3500])
