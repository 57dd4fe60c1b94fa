"""Soak test of the fast path on the bench scene with a different camera every step: SOAK_STEPS (300) steps, every 10th checked against the
operator path with (key, id) pairs and the radix pre-sort (all switches off) -- images bit-identical, binning outputs equal."""
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from gscodec_studio_amd import _step, rasterization  # noqa: E402
from gscodec_studio_amd import _wrapper as W  # noqa: E402
from gscodec_studio_amd._helper import sh_workload  # noqa: E402

dev = torch.device("cuda:0")
w = sh_workload(scene_grid=3, device=dev, n_cameras=16, camera_mode="jitter0")
P = {k: w[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")}
gen = torch.Generator(device=dev).manual_seed(0)


def render(vm, Ks):
    for p in P.values():
        p.grad = None
    rc, ra, meta = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["sh"], vm, Ks, 1920, 1080, sh_degree=3, packed=False)
    rc.sum().backward()
    return rc.detach(), ra.detach(), meta


bad = 0
for it in range(int(os.environ.get("SOAK_STEPS", "300"))):
    c = it % 16
    vm = w["viewmats"][c:c + 1].clone()
    vm[:, :3, 3] += 0.05 * torch.randn(1, 3, device=dev, generator=gen)  # a new pose every step
    Ks = w["Ks"][c:c + 1].contiguous()
    rc, ra, meta = render(vm, Ks)
    ids = meta["isect_ids"]
    assert bool((ids[1:] >= ids[:-1]).all()) and int(meta["tiles_per_gauss"].sum()) == ids.numel()
    assert bool(torch.isfinite(rc).all()) and all(bool(torch.isfinite(p.grad).all()) for p in P.values())
    if it % 10 == 0:
        prev = (_step.ENABLED, W._PACKED_PAIRS, dict(W._PRESORT))
        _step.ENABLED, W._PACKED_PAIRS = False, False
        W._PRESORT.update(on=False)
        try:
            rc2, ra2, meta2 = render(vm, Ks)
        finally:
            _step.ENABLED, W._PACKED_PAIRS = prev[0], prev[1]
            W._PRESORT.update(prev[2])
        ok = torch.equal(rc, rc2) and torch.equal(ra, ra2) and torch.equal(meta["isect_ids"], meta2["isect_ids"]) and \
            torch.equal(meta["flatten_ids"], meta2["flatten_ids"]) and torch.equal(meta["isect_offsets"], meta2["isect_offsets"])
        bad += 0 if ok else 1
        print(f"step {it}: camera {c} visible {int((meta['radii'] > 0).sum())} I {ids.numel()} {'identical' if ok else 'DIFFERENT'}", flush=True)
torch.cuda.synchronize()
print("soak:", "ok" if bad == 0 else f"{bad} mismatches")
sys.exit(1 if bad else 0)
