"""Where the GPU waits for the host inside one fast-path step (gs_step_* calls): pairs of events recorded right behind one
group of launches and right in front of the next -- the elapsed time between them is the idle time of the stream there."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from gscodec_studio_amd import _backend as B  # noqa: E402
from gscodec_studio_amd import rasterization  # noqa: E402
from gscodec_studio_amd._helper import sh_workload  # noqa: E402

dev = torch.device("cuda:0")
w = sh_workload(scene_grid=3, device=dev, camera_mode="jitter0")
params = {k: w[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")}
vm, Ks = w["viewmats"][:1].contiguous(), w["Ks"][:1].contiguous()
marks = []
orig = B.call


def call(name, *a):
    st = torch.cuda.current_stream()
    e0 = torch.cuda.Event(enable_timing=True)
    e0.record(st)
    r = orig(name, *a)
    e1 = torch.cuda.Event(enable_timing=True)
    e1.record(st)
    marks.append((name, e0, e1))
    return r


def step():
    for p in params.values():
        p.grad = None
    rc, ra, meta = rasterization(params["means"], params["quats"], params["scales"], params["opacities"], params["sh"], vm, Ks,
                                 1920, 1080, sh_degree=3, packed=False)
    rc.sum().backward()


for _ in range(10):
    step()
torch.cuda.synchronize()
B.call = call
n = 30
for _ in range(n):
    step()
torch.cuda.synchronize()
B.call = orig
per = len(marks) // n
names = [m[0] for m in marks[:per]]
import numpy as np
gaps = np.zeros((n, per))
durs = np.zeros((n, per))
for i in range(n):
    for j in range(per):
        nm, e0, e1 = marks[i * per + j]
        durs[i, j] = e0.elapsed_time(e1) * 1e3
        prev = marks[i * per + j - 1][2] if (i * per + j) > 0 else None
        gaps[i, j] = prev.elapsed_time(e0) * 1e3 if prev is not None else 0.0
print("call                          dur us (median)   idle-before us (median)   [events add ~6 us each]")
for j, nm in enumerate(names):
    print(f"{nm:30s} {np.median(durs[1:, j]):10.1f} {np.median(gaps[1:, j]):18.1f}")
print("step total (sum of medians):", round(float(np.median(durs[1:], 0).sum() + np.median(gaps[1:], 0).sum()), 1))
