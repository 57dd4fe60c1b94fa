"""Round-6 probe (verdict "next 3"): does running two cameras' steps CONCURRENTLY on one GPU -- two Python threads, two HIP streams,
each a full rasterization() forward + backward of its own camera over the same splats -- beat the serial C = 2 batch?  The binning
chain is ~165 us of small dependent launches that leave the VALUs idle; compositing is VALU-bound and leaves HBM idle: if the two
overlap, aggregate cameras/s goes up.  Prints cameras/s for: serial C = 1 twice, one C = 2 batch, two threads x C = 1."""
import os
import sys
import threading
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from gscodec_studio_amd import rasterization  # noqa: E402
from gscodec_studio_amd._helper import sh_workload  # noqa: E402

dev = torch.device("cuda:0")
w = sh_workload(scene_grid=3, device=dev, n_cameras=2, camera_mode="jitter0")
NAMES = ("means", "quats", "scales", "opacities", "sh")


def make(cams):
    P = [w[k].clone().requires_grad_(True) for k in NAMES]
    vm, Ks = w["viewmats"][cams].contiguous(), w["Ks"][cams].contiguous()

    def step():
        for p in P:
            p.grad = None
        rc, ra, meta = rasterization(*P, vm, Ks, w["width"], w["height"], sh_degree=3, packed=False)
        rc.sum().backward()

    return step


def timeit(fn, iters):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


ITERS = 200
s0, s1, s01 = make([0]), make([1]), make([0, 1])
t_serial = timeit(lambda: (s0(), s1()), ITERS // 2)
t_batch = timeit(s01, ITERS // 2)
print(f"serial: camera 0 then camera 1 on one stream: {t_serial:.3f} ms per pair  ({2 / t_serial * 1e3:.0f} cameras/s)")
print(f"one C = 2 batch:                              {t_batch:.3f} ms per pair  ({2 / t_batch * 1e3:.0f} cameras/s)")


def worker(step, stream, n, barrier, out, i):
    with torch.cuda.stream(stream):
        for _ in range(5):
            step()
        stream.synchronize()
        barrier.wait()
        t0 = time.perf_counter()
        for _ in range(n):
            step()
        stream.synchronize()
        out[i] = time.perf_counter() - t0


for rep in range(3):
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    bar = threading.Barrier(2)
    res = [0.0, 0.0]
    th = [threading.Thread(target=worker, args=(s, st, ITERS, bar, res, i)) for i, (s, st) in enumerate(zip((s0, s1), streams))]
    t0 = time.perf_counter()
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    wall = max(res)
    print(f"two threads, two streams (rep {rep}): {wall / ITERS * 1e3:.3f} ms per pair  ({2 * ITERS / wall:.0f} cameras/s), "
          f"x{t_serial / (wall / ITERS * 1e3):.3f} vs serial, x{t_batch / (wall / ITERS * 1e3):.3f} vs the C = 2 batch")
