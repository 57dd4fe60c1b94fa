import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
from test_gpu_dynamic_fused import _params, _P, KEYS
from util import T, N
from gscodec_studio_amd._wrapper import project_rows
from gscodec_studio_amd.dynamic import DynamicSlice, temporal_slice
fx, raw = _params(5000, seed=11)
vm, Ks = T(fx["viewmats"][:2]), T(fx["Ks"][:2])
res = []
inter = []
for fused in (True, False):
    P = _P(raw)
    ds = DynamicSlice(P["motion"], P["omega"], P["trbf_center"], P["trbf_scale"], 0.33)
    if fused:
        out = project_rows(P["means"], None, P["quats"], P["scales"], vm, Ks, fx["width"], fx["height"], P["opacities"], P["colors"], dynamic=ds)
    else:
        m, q, o, _ = temporal_slice(P["means"], P["motion"], P["quats"], P["omega"], P["opacities"], P["trbf_center"], P["trbf_scale"], 0.33)
        m.retain_grad(); q.retain_grad(); o.retain_grad()
        out = project_rows(m, None, q, P["scales"], vm, Ks, fx["width"], fx["height"], o, P["colors"])
    radii, means2d, depths, conics, opac, colors, rows = out
    g = torch.Generator(device="cuda:0").manual_seed(5)
    G = torch.randn(rows.shape, device=rows.device, generator=g) * (radii > 0)[..., None]
    torch.autograd.backward([means2d, conics, opac, colors, depths], [G[..., 0:2], G[..., 2:5], G[..., 5], G[..., 6:9], G[..., 9] * (radii > 0)])
    res.append({k: p.grad.clone() for k, p in P.items()})
for k in KEYS:
    a, b = res[0][k], res[1][k]
    d = (a != b)
    print(k, int(d.sum()), "of", a.numel(), "max rel", float(((a - b).abs() / (b.abs() + 1e-30))[d].max()) if d.any() else 0)
    if d.any():
        i = d.nonzero()[0]
        print("   at", i.tolist(), a[tuple(i)].item(), b[tuple(i)].item())
