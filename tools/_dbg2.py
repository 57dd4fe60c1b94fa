import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import numpy as np, torch
from util import N, rel_l2
from gscodec_studio_amd import rasterization
from gscodec_studio_amd._helper import DYNAMIC_KEYS, dynamic_workload
from gscodec_studio_amd.dynamic import temporal_slice
w = dynamic_workload(2_000_000, 1920, 1080, device="cuda:0")
W, H, vm, Ks, t = w["width"], w["height"], w["viewmats"], w["Ks"], 0.5
def run(fused, det=False):
    P0 = {k: w[k].clone().requires_grad_(True) for k in DYNAMIC_KEYS}
    from gscodec_studio_amd.compression_simulation import STGCompressionSimulation
    sim = STGCompressionSimulation(quantization_sim_type="round", entropy_steps={})
    P, _ = sim.simulate_compression(P0, step=1)
    scales, opac, tscale = torch.exp(P["scales"]), torch.sigmoid(P["opacities"]), torch.exp(P["trbf_scale"])
    if fused:
        rc, ra, meta = rasterization(P["means"], P["quats"], scales, opac, P["colors"], vm, Ks, W, H, packed=False, deterministic=det,
                                     dynamic=(P["motion"], P["omega"], P["trbf_center"], tscale, t))
    else:
        m_t, q_t, o_t, _ = temporal_slice(P["means"], P["motion"], P["quats"], P["omega"], opac, P["trbf_center"], tscale, t)
        rc, ra, meta = rasterization(m_t, q_t, scales, o_t, P["colors"], vm, Ks, W, H, packed=False, deterministic=det)
    rc.sum().backward()
    return {k: (p.grad.clone() if CLONE else p.grad) for k, p in P0.items() if p.grad is not None}
CLONE = False
for det in (False,):
    a = run(False, det); c = run(True, det); b = run(False, det); d = run(True, det)
    for k in ("means", "quats", "scales", "motion", "opacities", "colors"):
        print(det, k, "chain-chain %.2e  fused-fused %.2e  chain-fused %.2e" % (rel_l2(N(a[k]), N(b[k])), rel_l2(N(c[k]), N(d[k])), rel_l2(N(c[k]), N(a[k]))),
              "max|g| %.3g" % float(a[k].abs().max()))
