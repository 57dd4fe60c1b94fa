#!/usr/bin/env python
"""Golden vectors for the on-disk attribute quantizer, produced by RUNNING THE REFERENCE's
gsplat/compression/png_compression.py functions (_compress_png, _compress_png_kbit, _compress_png_16bit and the
matching _decompress_*) in the build container.  ``imageio`` is not installed; the functions only use it as a
lossless container, so an in-memory stand-in (imwrite keeps the array, imread returns it) is registered for the
duration of this script -- the arithmetic that is recorded is the reference's own.  Also pins oracle/codec_oracle.py.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_codec.py
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
STORE = {}
fake = types.ModuleType("imageio.v2")
fake.imwrite = lambda path, img: STORE.__setitem__(path, np.array(img, copy=True))
fake.imread = lambda path: STORE[path]
pkg = types.ModuleType("imageio")
pkg.v2 = fake
sys.modules["imageio"] = pkg
sys.modules["imageio.v2"] = fake
sys.path.insert(0, "/root/reference")
sys.path.insert(0, REPO)

import gsplat.compression.png_compression as P  # noqa: E402

from oracle import codec_oracle as CO  # noqa: E402


def main():
    rng = np.random.default_rng(5)
    side = 23
    n = side * side
    out = {"n_sidelen": side}
    cases = {
        "means16": (P._compress_png_16bit, P._decompress_png_16bit, (rng.normal(0, 1.5, (n, 3))).astype(np.float32), 16, {}),
        "scales8k": (P._compress_png_kbit, P._decompress_png_kbit, rng.uniform(-9, 1, (n, 3)).astype(np.float32), 8, {"quantization": 8}),
        "quats6k": (P._compress_png_kbit, P._decompress_png_kbit, rng.normal(0, 0.5, (n, 4)).astype(np.float32), 6, {"quantization": 6}),
        "opac8": (P._compress_png, P._decompress_png, rng.normal(0, 3, (n,)).astype(np.float32), 8, {}),
        "sh0_8k": (P._compress_png_kbit, P._decompress_png_kbit, rng.normal(0, 1, (n, 1, 3)).astype(np.float32), 8, {"quantization": 8}),
    }
    for name, (cfn, dfn, x, bits, kw) in cases.items():
        x[0] = x[1]  # exact ties in the data
        STORE.clear()
        meta = cfn("/mem", name, torch.from_numpy(x), n_sidelen=side, **kw)
        planes = [STORE[f"/mem/{name}_l.png"], STORE[f"/mem/{name}_u.png"]] if bits == 16 else [STORE[f"/mem/{name}.png"]]
        dec = dfn("/mem", name, meta).numpy()
        out[f"{name}.x"] = x
        out[f"{name}.bits"] = bits
        out[f"{name}.mins"] = np.asarray(meta["mins"], np.float32)
        out[f"{name}.maxs"] = np.asarray(meta["maxs"], np.float32)
        for i, p in enumerate(planes):
            out[f"{name}.plane{i}"] = p
        out[f"{name}.decoded"] = dec
        # pin the oracle: planes and decoded values bit-exact
        op, omin, omax = CO.quantize(x, side, bits)
        assert np.array_equal(omin, out[f"{name}.mins"]) and np.array_equal(omax, out[f"{name}.maxs"]), name
        for a, b in zip(op, planes):
            assert np.array_equal(a.squeeze(), b), name
        od = CO.dequantize([p.reshape(side, side, -1) for p in planes], meta["mins"], meta["maxs"], bits, x.shape)
        assert np.array_equal(od.view(np.uint32), dec.view(np.uint32)), name
        print(f"{name}: planes {[p.shape for p in planes]}, max |decoded - x| = {np.abs(dec - x).max():.4g} (oracle bit-exact)")
    path = os.path.join(HERE, "codec.npz")
    np.savez_compressed(path, **out)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


if __name__ == "__main__":
    main()
