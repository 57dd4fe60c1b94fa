"""Shared helpers for the test-suite."""
import os

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = os.path.join(HERE, "golden")
SH_C0 = 0.2820947917738781


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def garden(n=None, scale_mult=1.0):
    """Derived garden fixture (tests/golden/garden_small.npz) as a dict of numpy arrays."""
    fx = golden("garden_small.npz")
    d = {k: fx[k] for k in fx.files}
    if n is not None:
        for k in ("means", "scales", "quats", "opacities", "rgb"):
            d[k] = d[k][:n]
    d["scales"] = d["scales"] * scale_mult
    d["width"], d["height"] = int(d["width"]), int(d["height"])
    return d


def garden_sh(rgb, K=16, seed=0):
    """SH coefficients derived from the fixture colours (deterministic, legacy numpy RNG)."""
    rs = np.random.RandomState(seed)
    sh = np.zeros((rgb.shape[0], K, 3), np.float32)
    sh[:, 0] = (rgb - 0.5) / SH_C0
    if K > 1:
        sh[:, 1:] = rs.randn(rgb.shape[0], K - 1, 3).astype(np.float32) * 0.05
    return sh


def dev():
    return torch.device("cuda:0")


def T(a, requires_grad=False, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a), device=dev())
    if dtype is not None:
        t = t.to(dtype)
    if requires_grad:
        t.requires_grad_(True)
    return t


def N(t):
    return t.detach().cpu().numpy()


def assert_close(actual, expected, rtol, atol, what="", max_bad_frac=0.0):
    """|a - e| <= atol + rtol * |e| elementwise; optionally tolerate a tiny fraction of outliers."""
    a = np.asarray(actual, np.float64)
    e = np.asarray(expected, np.float64)
    assert a.shape == e.shape, f"{what}: shape {a.shape} vs {e.shape}"
    bad = np.abs(a - e) > (atol + rtol * np.abs(e))
    frac = bad.mean() if bad.size else 0.0
    if frac > max_bad_frac:
        i = np.unravel_index(np.argmax(np.abs(a - e) - rtol * np.abs(e)), a.shape) if a.size else ()
        raise AssertionError(
            f"{what}: {bad.sum()} / {bad.size} elements ({frac*100:.4f}%) outside rtol={rtol} atol={atol}; "
            f"worst at {i}: got {a[i]!r}, expected {e[i]!r}, max abs diff {np.abs(a-e).max():.3e}")


def rel_l2(a, e):
    a = np.asarray(a, np.float64)
    e = np.asarray(e, np.float64)
    return float(np.linalg.norm(a - e) / max(np.linalg.norm(e), 1e-30))


TUNING_DEFAULTS = {"raster_seg": 256, "raster_solo_min": 2048, "raster_xcd_fwd": 16, "raster_xcd_bwd": 16, "raster_order_fwd": 1}
# kernel routes a tuning value selects (all must give the reference's results; the fuzz tests run every one of them)
ROUTES = {
    "default": {},
    "all_solo": {"raster_solo_min": 1},        # every tile: four independent forward waves
    "no_solo": {"raster_solo_min": 0},         # every tile: cooperative staging, one barrier per batch
    "seg128": {"raster_seg": 128},             # shorter backward segments (more checkpoints)
    "unsegmented": {"raster_seg": 0},          # no checkpoints: one quadrant per wave walks the whole list backwards
    "xcd_identity": {"raster_xcd_fwd": 0, "raster_xcd_bwd": 0},
    "tile_order_off": {"raster_order_fwd": 0},  # forward workgroups in (XCD-remapped) tile order instead of longest list first
}


def set_tuning(**kv):
    """Tuning values for the rasterize calls that follow (they travel in each forward's gs_raster_plan; the library keeps
    no state)."""
    from gscodec_studio_amd import _wrapper as W

    W.set_raster_tuning(**kv)


class tuned:
    """``with tuned("all_solo"):`` -- run a block under one of ROUTES, restoring the defaults afterwards."""

    def __init__(self, route):
        self.kv = ROUTES[route] if isinstance(route, str) else dict(route)

    def __enter__(self):
        set_tuning(**self.kv)

    def __exit__(self, *a):
        set_tuning(**TUNING_DEFAULTS)
