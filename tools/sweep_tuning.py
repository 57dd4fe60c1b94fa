"""Re-tune the launch knobs of the compositing kernels on the bench workload, IN ONE PROCESS: the settings are interleaved
(round-robin over several rounds, median per setting), because separate bench.py processes differ by +-3 % from clock and
page-in state alone.  The values travel through set_raster_tuning -> gs_raster_plan (no library state)."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from gscodec_studio_amd import rasterization  # noqa: E402
from gscodec_studio_amd import _wrapper as W  # noqa: E402
from gscodec_studio_amd._helper import sh_workload  # noqa: E402

dev = torch.device("cuda:0")
w = sh_workload(scene_grid=3, device=dev)
params = {k: w[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")}
vm, Ks = w["viewmats"][:1].contiguous(), w["Ks"][:1].contiguous()


def step():
    for p in params.values():
        p.grad = None
    rc, ra, meta = rasterization(params["means"], params["quats"], params["scales"], params["opacities"], params["sh"], vm, Ks,
                                 1920, 1080, sh_degree=3, packed=False)
    rc.sum().backward()


def timed(n=40):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


settings = [{}]
if os.environ.get("SWEEP", "all") == "order":  # the knobs that interact with the forward's longest-first tile order
    settings.append({"raster_order_fwd": 0})
    for v in (512, 1024, 1536, 3072, 4096, 100000):
        settings.append({"raster_solo_min": v})
    settings.append({"raster_order_fwd": 0, "raster_solo_min": 4096})
else:
    for v in (192, 320, 384, 512):
        settings.append({"raster_seg": v})
    for v in (1024, 1536, 3072, 4096):
        settings.append({"raster_solo_min": v})
    for v in (8, 32, 64):
        settings.append({"raster_xcd_fwd": v})
    for v in (8, 32, 64):
        settings.append({"raster_xcd_bwd": v})
    settings.append({"raster_seg": 320, "raster_solo_min": 1024})
    settings.append({"raster_seg": 320, "raster_xcd_fwd": 32})
default = {"raster_seg": None, "raster_solo_min": None, "raster_xcd_fwd": None, "raster_xcd_bwd": None, "raster_order_fwd": None}
for _ in range(60):
    step()
res = {i: [] for i in range(len(settings))}
for rnd in range(int(os.environ.get("ROUNDS", "7"))):
    for i, s in enumerate(settings):
        W.set_raster_tuning(**default)
        W.set_raster_tuning(**s)
        for _ in range(3):
            step()
        res[i].append(timed())
W.set_raster_tuning(**default)
base = statistics.median(res[0])
for i, s in enumerate(settings):
    m = statistics.median(res[i])
    print(f"{m:.4f} ms/step ({(m / base - 1) * 100:+5.1f} %)  min {min(res[i]):.4f}  {s or 'defaults'}")
