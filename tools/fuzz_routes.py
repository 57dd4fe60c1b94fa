"""Offline fuzz (GPU box): random scenes through rasterization(), unpacked route (fused SH, means routed through the projection,
forward-cleared gradient rows) against the packed route (none of these) and against a second, repeated backward.
    python tools/fuzz_routes.py [n_cases]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from gscodec_studio_amd import rasterization  # noqa: E402


def rel(a, b):
    return float((a - b).norm() / b.norm().clamp(min=1e-20))


def main():
    n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 100
    dev = torch.device("cuda:0")
    worst = 0.0
    for case in range(n_cases):
        g = torch.Generator(device="cpu").manual_seed(case)
        N = int(torch.randint(1, 6000, (1,), generator=g))
        C = int(torch.randint(1, 4, (1,), generator=g))
        W = int(torch.randint(8, 300, (1,), generator=g))
        H = int(torch.randint(8, 200, (1,), generator=g))
        deg = int(torch.randint(0, 4, (1,), generator=g))
        mode = ["RGB", "RGB+D", "RGB+ED"][int(torch.randint(0, 3, (1,), generator=g))]
        means = (torch.rand(N, 3, generator=g) - 0.5) * torch.tensor([4.0, 4.0, 1.0])
        quats = torch.randn(N, 4, generator=g)
        scales = torch.rand(N, 3, generator=g) * 0.15 + 0.005
        opac = torch.rand(N, generator=g)
        sh = torch.randn(N, 16, 3, generator=g) * 0.3
        viewmats = torch.eye(4).repeat(C, 1, 1)
        viewmats[:, 2, 3] = 3.0 + torch.rand(C, generator=g)
        viewmats[:, 0, 3] = torch.randn(C, generator=g) * 0.3
        f = 0.8 * W
        Ks = torch.tensor([[f, 0, W / 2], [0, f, H / 2], [0, 0, 1.0]]).repeat(C, 1, 1)
        bg = torch.rand(C, 3, generator=g) if case % 3 == 0 else None
        P = [t.to(dev).requires_grad_(True) for t in (means, quats, scales, opac, sh)]
        kw = dict(sh_degree=deg, render_mode=mode, backgrounds=None if bg is None else bg.to(dev))
        rc, ra, _ = rasterization(*P, viewmats.to(dev), Ks.to(dev), W, H, packed=False, **kw)
        wgt = torch.rand(rc.shape, generator=g).to(dev)
        loss = (rc * wgt).sum() + ra.sum()
        g1 = torch.autograd.grad(loss, P, retain_graph=True, allow_unused=True)
        g2 = torch.autograd.grad(loss, P, allow_unused=True)
        rc_p, ra_p, _ = rasterization(*P, viewmats.to(dev), Ks.to(dev), W, H, packed=True, **kw)
        g3 = torch.autograd.grad((rc_p * wgt).sum() + ra_p.sum(), P, allow_unused=True)
        assert torch.isfinite(rc).all() and rel(rc_p, rc) < 1e-5 or float((rc_p - rc).abs().max()) < 1e-5, (case, "render")
        for a, b, c, name in zip(g1, g2, g3, ("means", "quats", "scales", "opacities", "sh")):
            if a is None:
                continue
            assert torch.isfinite(a).all(), (case, name)
            scale = float(a.abs().max()) + 1e-12
            e2, e3 = float((a - b).abs().max()) / scale, float((a - c).abs().max()) / scale
            worst = max(worst, e2, e3)
            assert e2 < 2e-3 and e3 < 2e-3, (case, name, e2, e3, N, C, W, H, deg, mode)
        if case % 20 == 0:
            print(f"case {case}: N={N} C={C} {W}x{H} deg={deg} {mode} ok, worst so far {worst:.2e}", flush=True)
    print(f"{n_cases} cases ok, worst max-norm difference {worst:.2e}")


if __name__ == "__main__":
    main()
