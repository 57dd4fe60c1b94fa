"""C = 8 cameras on one GPU, forward + backward (the workload of tools/bench_multicam.py's last line), for profiling."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from gscodec_studio_amd import rasterization
from gscodec_studio_amd._helper import sh_workload

import gc  # noqa: E402

# a full collection over the ~10^5 objects torch's import leaves behind takes 30-50 ms and lands in the middle of a timed loop
# (one 33 ms call in 30: a "2.1 ms" forward that is 0.44): park them in the permanent generation
gc.collect()
gc.freeze()
C = int(os.environ.get("C", "8"))
w = sh_workload(scene_grid=3, device="cuda:0", n_cameras=C, camera_mode="jitter0")
P = [w[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")]
def step():
    for p in P: p.grad = None
    rc, ra, meta = rasterization(*P, w["viewmats"], w["Ks"], w["width"], w["height"], sh_degree=3, packed=False)
    rc.sum().backward()
for _ in range(3): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10 * 1e3
print(f"C={C}: {dt:.3f} ms/step, {dt / C:.3f} ms per camera")
