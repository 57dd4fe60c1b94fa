"""Feature rendering at BASELINE config 2's size: 1,006,065 gaussians, one 1080p camera, D post-activation colour channels
(no SH) -- D = 9 is what the reference's spacetime trainer renders every step (examples/simple_trainer_STG.py:531-551:
colors = cat(feature_color, feature_dir, t * feature_time)), D = 32 the reference's published feature-map row
(docs/source/tests/profile.rst:76-93).  A step = rasterization() forward + backward of sum(render); also forward alone.
usage: python tools/bench_channels.py [D ...]      (default 3 9 16 32)"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gscodec_studio_amd import rasterization  # noqa: E402
from gscodec_studio_amd._helper import sh_workload  # noqa: E402

import gc  # noqa: E402

# a full collection over the ~10^5 objects torch's import leaves behind takes 30-50 ms and lands in the middle of a timed loop
# (one 33 ms call in 30: a "2.1 ms" forward that is 0.44): park them in the permanent generation
gc.collect()
gc.freeze()


def run(D, steps=30, grid=3):
    dev = torch.device("cuda")
    w = sh_workload(scene_grid=grid, width=1920, height=1080, n_cameras=1, sh_degree=0, device=dev)
    N = w["N"]
    g = torch.Generator(device="cpu").manual_seed(D)
    colors = torch.rand(N, D, generator=g).to(dev)
    ps = {k: w[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")}
    colors.requires_grad_(True)

    def step(bwd=True):
        for p in list(ps.values()) + [colors]:
            p.grad = None
        rc, ra, meta = rasterization(ps["means"], ps["quats"], ps["scales"], ps["opacities"], colors, w["viewmats"], w["Ks"], 1920, 1080,
                                     packed=False)
        if bwd:
            rc.sum().backward()
        return meta

    def timed(bwd):
        for _ in range(5):
            step(bwd)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        calls = []
        for _ in range(steps):
            t1 = time.perf_counter()
            meta = step(bwd)
            calls.append((time.perf_counter() - t1) * 1e3)
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / steps * 1e3
        if os.environ.get("GS_BENCH_CALLS"):
            print(f"    host ms per call ({'fwd+bwd' if bwd else 'fwd'}):", " ".join(f"{c:.2f}" for c in calls), flush=True)
        return t, meta

    with torch.no_grad():
        t_f, _ = timed(False)
    t_fb, meta = timed(True)
    print(f"D = {D:2d}  N = {N}  I = {meta['flatten_ids'].numel()}  fwd+bwd {t_fb:7.3f} ms/step = {N / t_fb / 1e3:7.1f} Msplats/s   "
          f"fwd only (no_grad) {t_f:7.3f} ms", flush=True)
    return t_fb


if __name__ == "__main__":
    Ds = [int(a) for a in sys.argv[1:]] or [3, 9, 16, 32]
    base = None
    for D in Ds:
        torch.cuda.empty_cache()
        t = run(D)
        if D == 3:
            base = t
        elif base:
            print(f"        -> {t / base:.2f}x the D = 3 step", flush=True)
