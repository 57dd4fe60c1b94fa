"""The reference's own rasterizer micro-benchmark (profiling/main.py:28-151) restated on this package, for a like-for-like line
next to docs/source/tests/profile.rst (TITAN RTX): 1080p, 3 post-activation colour channels (no SH), near 0.01, far 100,
radius_clip 3, batch 1; forward = `repeats` calls of rasterization() between synchronisations after 5 warm-ups, backward =
`repeats` x loss.backward(retain_graph=True) of loss = render_colors.sum(); Msplats/s = N / (1/FPS_fwd + 1/FPS_bwd).
`--channels D` is the reference's "more channels" knob (profiling/main.py:69-70: colors[:, :1].repeat(1, channels); profile.rst:76-93
publishes the 32-channel rows at scene_grid 1).
usage: python tools/bench_profile_protocol.py [--channels D] [--unpacked-only] [scene_grid ...]"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gscodec_studio_amd import rasterization  # noqa: E402
from gscodec_studio_amd._helper import load_test_data, rescale_intrinsics  # noqa: E402

import gc  # noqa: E402

# a full collection over the ~10^5 objects torch's import leaves behind takes 30-50 ms and lands in the middle of a timed loop
# (one 33 ms call in 30: a "2.1 ms" forward that is 0.44): park them in the permanent generation
gc.collect()
gc.freeze()


def timeit(repeats, f, *args, **kw):
    for _ in range(5):
        out = f(*args, **kw)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(repeats):
        out = f(*args, **kw)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / repeats, out


def main(grid, packed, sparse_grad=False, repeats=30, channels=3):
    dev = torch.device("cuda")
    means, quats, scales, opacities, colors, viewmats, Ks, w0, h0 = load_test_data(device="cpu", scene_grid=grid)
    W, H = 1920, 1080
    Ks = rescale_intrinsics(Ks, w0, h0, W, H)
    viewmats, Ks = viewmats[:1].to(dev), Ks[:1].to(dev)
    colors = colors[:, :1].repeat(1, channels)
    ps = [t.to(dev).contiguous().requires_grad_(True) for t in (means, quats, scales, opacities, colors)]
    t_fwd, out = timeit(repeats, rasterization, *ps, viewmats, Ks, W, H, packed=packed, near_plane=0.01, far_plane=100.0,
                        radius_clip=3.0, sparse_grad=sparse_grad)
    loss = out[0].sum()

    def backward():
        loss.backward(retain_graph=True)
        for v in ps:
            v.grad = None

    t_bwd, _ = timeit(repeats, backward)
    N = ps[0].shape[0]
    print(f"grid {grid:2d}  N = {N:>11,d}  channels={channels:2d}  packed={packed!s:5s} sparse_grad={sparse_grad!s:5s}  FPS fwd {1 / t_fwd:8.1f}  bwd {1 / t_bwd:8.1f}  "
          f"-> {N / (t_fwd + t_bwd) / 1e6:8.1f} Msplats/s fwd+bwd ({N / t_fwd / 1e6:8.1f} fwd only)", flush=True)


if __name__ == "__main__":
    argv = sys.argv[1:]
    ch, unpacked_only = 3, False
    if "--channels" in argv:
        i = argv.index("--channels")
        ch = int(argv[i + 1])
        del argv[i:i + 2]
    if "--unpacked-only" in argv:
        argv.remove("--unpacked-only")
        unpacked_only = True
    grids = [int(a) for a in argv] or [5]
    for g in grids:
        main(g, packed=False, channels=ch)
        if not unpacked_only:
            main(g, packed=True, channels=ch)
            main(g, packed=True, sparse_grad=True, channels=ch)
