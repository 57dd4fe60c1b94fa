#!/usr/bin/env python
"""float64 evaluation of the REFERENCE's projection (gsplat/cuda/_torch_impl.py:_fully_fused_projection +
_quat_scale_to_covar_preci, imported in the build container) on the inputs and cotangents of tests/golden/projection.npz,
with the visibility of the fp32 run (its radii): the gradients the fp32 golden vectors approximate, to ~1e-16.  The GPU test
(tests/test_gpu_ops.py) holds the HIP backward to 1e-4 relative L2 of THESE, i.e. the north-star tolerance is asserted
against the reference itself and not only against the oracle chain.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_projection_f64.py

Writes tests/golden/projection_f64.npz (gradients only, float64; data, no reference source) and reports how far the fp32
golden gradients and the fp32 oracle are from them.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.modules["_gridencoder"] = types.ModuleType("_gridencoder")
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

from gsplat.cuda import _torch_impl as T  # noqa: E402


def rel_l2(a, b):
    return float(np.linalg.norm(np.asarray(a, np.float64) - b) / max(np.linalg.norm(b), 1e-300))


def main():
    gd = np.load(os.path.join(HERE, "projection.npz"))
    W, H = int(gd["width"]), int(gd["height"])
    d = lambda k: torch.tensor(gd[k], dtype=torch.float64)  # noqa: E731
    out = {}
    for model in ["pinhole", "ortho", "fisheye"]:
        for comp in [False, True]:
            tag = f"{model}_{int(comp)}"
            mm, qq, ss, vv = (d(k).requires_grad_(True) for k in ("means", "quats", "scales", "viewmats"))
            covars, _ = T._quat_scale_to_covar_preci(qq, ss, triu=False)
            radii, means2d, depths, conics, comps = T._fully_fused_projection(
                mm, covars, vv, d("Ks"), W, H, calc_compensations=comp, camera_model=model)
            valid = torch.tensor(gd[f"{tag}_radii"] > 0)  # the fp32 run's visibility: same terms in the loss
            assert bool((radii[valid] > 0).all())
            loss = (means2d * d("v_means2d") * valid[..., None]).sum() + (depths * d("v_depths") * valid).sum() + \
                   (conics * d("v_conics") * valid[..., None]).sum()
            if comp:
                loss = loss + (comps * d("v_comp") * valid).sum()
            grads = torch.autograd.grad(loss, (mm, qq, ss, vv))
            for name, g in zip(("v_means", "v_quats", "v_scales", "v_viewmats"), grads):
                out[f"{tag}_{name}"] = g.numpy()
                print(f"[{tag}] {name}: fp32 reference golden is {rel_l2(gd[f'{tag}_{name}'], g.numpy()):.2e} rel. L2 from float64")
    path = os.path.join(HERE, "projection_f64.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
