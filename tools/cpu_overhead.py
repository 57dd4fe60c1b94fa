"""Host-side cost of one rasterization() forward + backward: a scene so small that every kernel sits at its launch floor, so
the wall time per step IS the host time (Python + torch dispatch + ctypes + launches).  Prints it, and cProfile's top entries."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from gscodec_studio_amd import rasterization  # noqa: E402
from gscodec_studio_amd._helper import sh_workload  # noqa: E402

dev = torch.device("cuda:0")
DIST = os.environ.get("DIST", "0") == "1"  # gaussian-sharded mode at world 1, collectives forced through RCCL
if DIST:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29733")
    os.environ["GS_DIST_FORCE_COLLECTIVES"] = "1"
    torch.cuda.set_device(dev)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1)
w = sh_workload(scene_grid=1, device=dev)
n = int(os.environ.get("N", "3000"))
params = {k: w[k][:n].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")}
vm, Ks = w["viewmats"][:1].contiguous(), w["Ks"][:1].contiguous()


def step():
    for p in params.values():
        p.grad = None
    rc, ra, meta = rasterization(params["means"], params["quats"], params["scales"], params["opacities"], params["sh"], vm, Ks,
                                 1920, 1080, sh_degree=3, packed=False, distributed=DIST)
    rc.sum().backward()


for _ in range(20):
    step()
torch.cuda.synchronize()
K = 200
if os.environ.get("AB", "0") == "1" and not DIST:
    # step driver on / off interleaved in ONE process (separate processes land on differently loaded CPUs of the pod)
    from gscodec_studio_amd import _step

    res = {True: [], False: []}
    for rnd in range(9):
        for on in (True, False):
            _step.ENABLED = on
            for _ in range(20):
                step()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(K):
                step()
            torch.cuda.synchronize()
            res[on].append(1e3 * (time.perf_counter() - t0) / K)
    for on in (True, False):
        r = sorted(res[on])
        print(f"host-bound step, step driver {'on ' if on else 'off'}: min {r[0]:.3f} median {r[len(r) // 2]:.3f} max {r[-1]:.3f} ms "
              f"per forward + backward ({n} splats, 9 interleaved rounds of {K} steps)")
    sys.exit(0)
runs = []
for _ in range(7):  # (the pod's CPUs are shared: the minimum over a few repeats is the host's own cost, the median shows the noise)
    t0 = time.perf_counter()
    for _ in range(K):
        step()
    torch.cuda.synchronize()
    runs.append(1e3 * (time.perf_counter() - t0) / K)
runs.sort()
print(f"host-bound step: {runs[0]:.3f} ms per forward + backward ({n} splats; min of 7 x {K} steps, median {runs[3]:.3f}, max {runs[-1]:.3f})")
pr = cProfile.Profile()
pr.enable()
for _ in range(K):
    step()
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(45)
print(s.getvalue()[:9000])
