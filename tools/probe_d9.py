"""Why `bench_channels.py 3 9` reads ~2 ms for the D = 9 forward-only phase when `bench_channels.py 9` reads 0.44: per-call host
times of the same sequence (no synchronisation inside the loop, as in the tool)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gscodec_studio_amd import rasterization
from gscodec_studio_amd._helper import sh_workload

def run(D, steps=30):
    dev = torch.device("cuda")
    w = sh_workload(scene_grid=3, width=1920, height=1080, n_cameras=1, sh_degree=0, device=dev)
    N = w["N"]
    colors = torch.rand(N, D, device=dev).requires_grad_(True)
    ps = {k: w[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities")}
    def step(bwd):
        for p in list(ps.values()) + [colors]:
            p.grad = None
        rc, ra, meta = rasterization(ps["means"], ps["quats"], ps["scales"], ps["opacities"], colors, w["viewmats"], w["Ks"], 1920, 1080, packed=False)
        if bwd:
            rc.sum().backward()
    def timed(bwd, tag):
        ts = []
        torch.cuda.synchronize()
        t00 = time.perf_counter()
        for _ in range(5 + steps):
            t0 = time.perf_counter(); step(bwd); ts.append((time.perf_counter() - t0) * 1e3)
        torch.cuda.synchronize()
        tot = (time.perf_counter() - t00) * 1e3
        print(f"D={D} {tag}: total {tot:.1f} ms over {5 + steps} calls; host per call:", " ".join(f"{t:.2f}" for t in ts), flush=True)
    with torch.no_grad():
        timed(False, "fwd")
    timed(True, "fwd+bwd")

for D in [int(a) for a in sys.argv[1:]] or [3, 9]:
    torch.cuda.empty_cache()
    run(D)
