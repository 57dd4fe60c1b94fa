#!/usr/bin/env python
"""Golden vectors for the DECODE PIPELINE of the on-disk format: a compressed directory in the reference's own layout
(PNG planes written by the reference's _compress_png* functions through an in-memory stand-in for ``imageio``, a K-means
``shN.npz`` in the reference's format, meta.json) is decoded by the reference's ``PngCompression.decompress``
(gsplat/compression/png_compression.py:132-152), run in the build container; planes, metadata and the decoded splats are
recorded.  tests/test_gpu_codec.py feeds the planes to ``decode_to_rasterizer_inputs`` / ``kmeans_decode`` and requires the
decoded raw parameters bit for bit (the fused activations to 1e-6).  ``torchpq`` and ``plas`` are not installed, so the
codebook here is a small hand-made one (the decoder does not care where it came from) and no PLAS sort runs.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_codec_pipeline.py
"""
import json
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.modules["_gridencoder"] = types.ModuleType("_gridencoder")
STORE = {}
fake = types.ModuleType("imageio.v2")
fake.imwrite = lambda path, img: STORE.__setitem__(os.path.basename(path), np.array(img, copy=True))
fake.imread = lambda path: STORE[os.path.basename(path)]
pkg = types.ModuleType("imageio")
pkg.v2 = fake
sys.modules["imageio"] = pkg
sys.modules["imageio.v2"] = fake
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

import gsplat.compression.png_compression as P  # noqa: E402
from gsplat.utils import log_transform  # noqa: E402


def main():
    rng = np.random.default_rng(11)
    side = 19
    n = side * side
    splats = {
        "means": torch.from_numpy(rng.normal(0, 4.0, (n, 3)).astype(np.float32)),
        "scales": torch.from_numpy(rng.uniform(-9, 1, (n, 3)).astype(np.float32)),
        "quats": torch.nn.functional.normalize(torch.from_numpy(rng.normal(0, 1, (n, 4)).astype(np.float32)), dim=-1),
        "opacities": torch.from_numpy(rng.normal(0, 3, (n,)).astype(np.float32)),
        "sh0": torch.from_numpy(rng.normal(0, 1, (n, 1, 3)).astype(np.float32)),
    }
    comp = P.PngCompression(use_sort=False, verbose=False)
    out = {"n_sidelen": side}
    with tempfile.TemporaryDirectory() as d:
        meta = {}
        pre = dict(splats)
        pre["means"] = log_transform(pre["means"])  # as PngCompression.compress does (png_compression.py:100)
        for name, value in pre.items():
            meta[name] = comp._get_compress_fn(name)(d, name, value, n_sidelen=side, verbose=False)
        # shN through the fork's MASKED K-means container (png_compression.py:523-640): a bit-packed mask of the splats
        # that carry higher bands, a hand-made codebook of 37 rows x 45, uint16 labels for the masked splats only,
        # scalar min / max
        k, w = 37, 45
        mask = rng.random(n) < 0.7
        bits = np.packbits(mask)[: (n + 7) // 8]
        bits.tofile(os.path.join(d, "mask.bin"))
        cq = rng.integers(0, 256, (k, w)).astype(np.uint8)
        labels = rng.integers(0, k, int(mask.sum())).astype(np.uint16)
        np.savez_compressed(os.path.join(d, "shN.npz"), centroids=cq, labels=labels)
        meta["shN"] = {"shape": [n, 15, 3], "dtype": "float32", "mins": -0.3712, "maxs": 0.4519, "quantization": 8,
                       "mask_bits": n, "mask_byte": (n + 7) // 8}
        with open(os.path.join(d, "meta.json"), "w") as f:
            json.dump(meta, f)
        dec = comp.decompress(d)
    for name in ("means", "scales", "quats", "opacities", "sh0"):
        m = meta[name]
        if name == "means":
            out["means.plane0"], out["means.plane1"] = STORE["means_l.png"], STORE["means_u.png"]
        else:
            out[f"{name}.plane0"] = STORE[f"{name}.png"]
        out[f"{name}.mins"] = np.asarray(m["mins"], np.float32)
        out[f"{name}.maxs"] = np.asarray(m["maxs"], np.float32)
        out[f"{name}.bits"] = 16 if name == "means" else int(m.get("quantization", 8))
        out[f"{name}.shape"] = np.asarray(m["shape"])
        out[f"{name}.decoded"] = dec[name].numpy()
    out["shN.centroids"], out["shN.labels"], out["shN.mask"] = cq, labels, mask
    out["shN.mins"], out["shN.maxs"] = np.float32(meta["shN"]["mins"]), np.float32(meta["shN"]["maxs"])
    out["shN.decoded"] = dec["shN"].numpy()
    out["input.means"] = splats["means"].numpy()
    path = os.path.join(HERE, "codec_pipeline.npz")
    np.savez_compressed(path, **out)
    err = np.abs(out["means.decoded"] - out["input.means"]).max()
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB; attributes {list(dec)}; max |means decoded - input| = {err:.3g}")


if __name__ == "__main__":
    main()
