#!/usr/bin/env python
"""Exporter of REFERENCE-CUDA golden vectors for the compositing stage (R5 / R5b) -- the one stage whose parity cannot be
pinned inside the build container (the reference's `rasterize_to_pixels` needs nvcc, glm and a CUDA GPU).

Run it ONCE on any machine that has a CUDA GPU and the reference installed (`pip install -e /path/to/GSCodec_Studio`, or
`PYTHONPATH=/path/to/GSCodec_Studio`), from the root of this repository:

    python tests/golden/export_raster_cuda.py            # writes tests/golden/raster_cuda.npz  (~6 MB)

and commit the file.  `tests/test_gpu_cuda_golden.py` loads it when present (and skips, saying so, when it is not) and
compares the HIP kernels with it: tile / bin indices bit-exact, render and gradients within 1e-4 relative.

Only DATA is written: the committed fixture inputs (tests/golden/garden_small.npz), the reference's intermediates
(projection, binning), its outputs and its gradients for seeded upstream gradients.  No reference source is copied.
"""
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "raster_cuda.npz")
SH_C0 = 0.2820947917738781


def main():
    assert torch.cuda.is_available(), "needs a CUDA GPU"
    import gsplat  # the REFERENCE (JasonLSC/GSCodec_Studio), not this repository
    from gsplat.cuda._wrapper import fully_fused_projection, isect_offset_encode, isect_tiles, rasterize_to_pixels
    from gsplat.rendering import rasterization

    assert hasattr(gsplat, "compression_simulation"), "this is not the GSCodec_Studio fork of gsplat"
    dev = torch.device("cuda:0")
    fx = np.load(os.path.join(HERE, "garden_small.npz"))
    n, cams, scale_mult = 4000, 2, 6.0
    W, H = int(fx["width"]), int(fx["height"])
    t = lambda a, g=False: torch.tensor(np.ascontiguousarray(a), device=dev, requires_grad=g)  # noqa: E731
    means, quats, scales = fx["means"][:n], fx["quats"][:n], fx["scales"][:n] * scale_mult
    opac1 = np.clip(fx["opacities"][:n] * 3.0, 0, 1).astype(np.float32)  # saturating: early termination + alpha clamp
    vm, Ks = fx["viewmats"][:cams], fx["Ks"][:cams]
    rs = np.random.RandomState(0)
    colors = rs.rand(cams, n, 3).astype(np.float32)
    bg = rs.rand(cams, 3).astype(np.float32)
    out = dict(n=n, cams=cams, scale_mult=scale_mult, width=W, height=H, colors=colors, backgrounds=bg, opacities_n=opac1)

    # ---- stage level: projection -> binning -> rasterize_to_pixels forward + backward
    with torch.no_grad():
        radii, means2d, depths, conics, _ = fully_fused_projection(t(means), None, t(quats), t(scales), t(vm), t(Ks), W, H,
                                                                   packed=False)
        tw, th = math.ceil(W / 16), math.ceil(H / 16)
        tpg, isect_ids, flatten_ids = isect_tiles(means2d, radii, depths, 16, tw, th)
        offsets = isect_offset_encode(isect_ids, cams, tw, th)
    out.update(radii=radii.cpu().numpy(), means2d=means2d.cpu().numpy(), depths=depths.cpu().numpy(), conics=conics.cpu().numpy(),
               tiles_per_gauss=tpg.cpu().numpy(), isect_ids=isect_ids.cpu().numpy(), flatten_ids=flatten_ids.cpu().numpy(),
               isect_offsets=offsets.cpu().numpy())
    opac = t(np.broadcast_to(opac1[None], (cams, n)).copy(), True)
    m2, cn, col, bg_t = means2d.clone().requires_grad_(True), conics.clone().requires_grad_(True), t(colors, True), t(bg, True)
    rc, ra = rasterize_to_pixels(m2, cn, col, opac, W, H, 16, offsets, flatten_ids, backgrounds=bg_t, absgrad=True)
    v_rc = rs.randn(cams, H, W, 3).astype(np.float32)
    v_ra = rs.randn(cams, H, W, 1).astype(np.float32)
    ((rc * t(v_rc)).sum() + (ra * t(v_ra)).sum()).backward()
    out.update(render_colors=rc.detach().cpu().numpy(), render_alphas=ra.detach().cpu().numpy(), v_render_colors=v_rc,
               v_render_alphas=v_ra, v_means2d=m2.grad.cpu().numpy(), v_conics=cn.grad.cpu().numpy(), v_colors=col.grad.cpu().numpy(),
               v_opacities=opac.grad.cpu().numpy(), v_backgrounds=bg_t.grad.cpu().numpy(), absgrad=m2.absgrad.cpu().numpy())

    # ---- API level: rasterization() with SH degree 3, forward + backward to the splat parameters
    sh = np.zeros((n, 16, 3), np.float32)
    sh[:, 0] = (fx["rgb"][:n] - 0.5) / SH_C0
    sh[:, 1:] = np.random.RandomState(0).randn(n, 15, 3).astype(np.float32) * 0.05
    P = dict(means=t(means, True), quats=t(quats, True), scales=t(scales, True), opacities=t(opac1, True), sh=t(sh, True))
    rc2, ra2, meta = rasterization(P["means"], P["quats"], P["scales"], P["opacities"], P["sh"], t(vm), t(Ks), W, H, sh_degree=3,
                                   packed=False)
    v2 = np.random.RandomState(1).randn(cams, H, W, 3).astype(np.float32)
    (rc2 * t(v2)).sum().backward()
    out.update(api_sh=sh, api_render_colors=rc2.detach().cpu().numpy(), api_render_alphas=ra2.detach().cpu().numpy(),
               api_v_render_colors=v2, api_radii=meta["radii"].cpu().numpy(), api_flatten_ids=meta["flatten_ids"].cpu().numpy(),
               api_isect_offsets=meta["isect_offsets"].cpu().numpy(),
               **{f"api_grad_{k}": p.grad.cpu().numpy() for k, p in P.items()})
    out["gsplat_version"] = np.array(getattr(gsplat, "__version__", "?"))
    out["device"] = np.array(torch.cuda.get_device_name(0))
    np.savez_compressed(OUT, **out)
    print(f"wrote {OUT} ({os.path.getsize(OUT) / 1e6:.1f} MB) from gsplat {out['gsplat_version']} on {out['device']}")


if __name__ == "__main__":
    sys.exit(main())
