import torch, time, sys
sys.path.insert(0, "/root/repo")
from gscodec_studio_amd import rasterization
from gscodec_studio_amd._helper import sh_workload

import gc  # noqa: E402

# a full collection over the ~10^5 objects torch's import leaves behind takes 30-50 ms and lands in the middle of a timed loop
# (one 33 ms call in 30: a "2.1 ms" forward that is 0.44): park them in the permanent generation
gc.collect()
gc.freeze()
w = sh_workload(scene_grid=3, device="cuda:0")
args = (w["means"], w["quats"], w["scales"], w["opacities"], w["sh"], w["viewmats"], w["Ks"], w["width"], w["height"])
def run(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        rasterization(*args, sh_degree=3, packed=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    run(5); print("inference ms/frame (no grad):", run(30))
P = [a.clone().requires_grad_(True) if i < 5 else a for i, a in enumerate(args)]
def run2(n):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n):
        rasterization(*P, sh_degree=3, packed=False)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
run2(5); print("forward only with grad-enabled inputs ms:", run2(30))
