"""Host-side timeline of one fast-path forward (tools/step_gaps.py looks at the GPU's idle time; this at where the HOST spends it):
perf_counter stamps at every native call's entry / exit and around the sentinel wait, median over steps, relative to the start of
rasterization()."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

from gscodec_studio_amd import _backend as B  # noqa: E402
from gscodec_studio_amd import _wrapper as W  # noqa: E402
from gscodec_studio_amd import rasterization  # noqa: E402
from gscodec_studio_amd._helper import sh_workload  # noqa: E402

dev = torch.device("cuda:0")
w = sh_workload(scene_grid=3, device=dev, camera_mode="jitter0")
params = {k: w[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")}
vm, Ks = w["viewmats"][:1].contiguous(), w["Ks"][:1].contiguous()
log = []
orig_call, orig_wait = B.call, W._wait_event


def call(name, *a):
    t0 = time.perf_counter()
    r = orig_call(name, *a)
    log.append((name, t0, time.perf_counter()))
    return r


def wait(ev):
    t0 = time.perf_counter()
    orig_wait(ev)
    log.append(("wait(block sums)", t0, time.perf_counter()))


def step(trace):
    for p in params.values():
        p.grad = None
    t0 = time.perf_counter()
    rc, ra, meta = rasterization(params["means"], params["quats"], params["scales"], params["opacities"], params["sh"], vm, Ks, 1920, 1080,
                                 sh_degree=3, packed=False)
    t1 = time.perf_counter()
    rc.sum().backward()
    return t0, t1


for _ in range(20):
    step(False)
torch.cuda.synchronize()
B.call, W._wait_event = call, wait
rows = []
for _ in range(200):
    del log[:]
    t0, t1 = step(True)
    fwd = [(n, a - t0, b - t0) for n, a, b in log if a < t1]
    rows.append(fwd + [("rasterization() returns", t1 - t0, t1 - t0)])
B.call, W._wait_event = orig_call, orig_wait
names = [n for n, _, _ in rows[0]]
print(f"{'host event':34s} enter us   leave us   (medians over {len(rows)} steps, t = 0 at the call of rasterization())")
for j, n in enumerate(names):
    a = np.median([r[j][1] for r in rows if len(r) == len(names)]) * 1e6
    b = np.median([r[j][2] for r in rows if len(r) == len(names)]) * 1e6
    print(f"{n:34s} {a:8.1f} {b:10.1f}")
