"""Timing of the fused temporal slicing at config-5 size (N = 2M dynamic splats) against the eager-torch chain of
the reference trainer's formulas on the same GPU.  Prints one JSON line."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from gscodec_studio_amd.dynamic import temporal_slice  # noqa: E402

import gc  # noqa: E402

# a full collection over the ~10^5 objects torch's import leaves behind takes 30-50 ms and lands in the middle of a timed loop
# (one 33 ms call in 30: a "2.1 ms" forward that is 0.44): park them in the permanent generation
gc.collect()
gc.freeze()


def eager(means, motion, quats, omega, opac, c, s, t):
    tau = t - c
    trbf = torch.exp(-1 * (tau / (2 ** 0.5 * s)).pow(2))
    o = opac * trbf.squeeze()
    tp = tau.detach()
    m = means + motion[:, 0:3] * tp + motion[:, 3:6] * tp * tp + motion[:, 6:9] * tp * tp * tp
    q = torch.nn.functional.normalize(quats + tp * omega)
    return m, q, o


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / iters * 1e3


def main():
    n = 2_000_000
    g = lambda *sh: torch.randn(*sh, device="cuda", requires_grad=True)  # noqa: E731
    means, motion, quats, omega = g(n, 3), g(n, 9), g(n, 4), g(n, 4)
    opac = torch.rand(n, device="cuda", requires_grad=True)
    c = torch.rand(n, 1, device="cuda", requires_grad=True)
    s = torch.rand(n, 1, device="cuda").add(0.1).requires_grad_(True)
    ins = (means, motion, quats, omega, opac, c, s)

    def run(f):
        for p in ins:
            p.grad = None
        out = f(*ins, 0.4)
        (out[0].sum() + out[1].sum() + out[2].sum()).backward()

    with torch.no_grad():
        tf = timeit(lambda: temporal_slice(*ins, 0.4))
        te = timeit(lambda: eager(*ins, 0.4))
    tfb = timeit(lambda: run(lambda *a: temporal_slice(*a)[:3]))
    teb = timeit(lambda: run(eager))
    print(json.dumps({"workload": "temporal slice, N=2000000", "fused_fwd_ms": round(tf, 4), "eager_fwd_ms": round(te, 4),
                      "fused_fwd_bwd_ms": round(tfb, 4), "eager_fwd_bwd_ms": round(teb, 4),
                      "fwd_GBps": round(128 * n / tf / 1e6, 1)}))


if __name__ == "__main__":
    main()
