"""CPU oracle for the on-disk attribute quantizer -- TEST INFRASTRUCTURE ONLY.

numpy restatement of the arithmetic of _compress_png / _compress_png_kbit / _compress_png_16bit and their
_decompress_* counterparts (gsplat/compression/png_compression.py:166-389), without the PNG container.
Pinned by tests/golden/make_golden_codec.py, which runs the reference functions themselves with an in-memory
stand-in for ``imageio.v2`` (imwrite/imread keep the arrays in a dict; a PNG round trip is lossless)."""
import numpy as np


def quantize(params: np.ndarray, n_sidelen: int, bits: int):
    """fp32 normalisation, round half to even.  Returns (planes, mins, maxs); k-bit images are shifted left."""
    grid = params.reshape(n_sidelen, n_sidelen, -1).astype(np.float32)
    mins = grid.min(axis=(0, 1))
    maxs = grid.max(axis=(0, 1))
    with np.errstate(invalid="ignore", divide="ignore"):
        norm = (grid - mins) / (maxs - mins)
        lvl = np.round(norm * np.float32(2**bits - 1))
    lvl = np.nan_to_num(lvl, nan=0.0)
    if bits == 16:
        img = lvl.astype(np.uint16)
        return [(img & 0xFF).astype(np.uint8), ((img >> 8) & 0xFF).astype(np.uint8)], mins, maxs
    img = lvl.astype(np.uint8) << (8 - bits)
    return [img.astype(np.uint8)], mins, maxs


def dequantize(planes, mins, maxs, bits: int, shape):
    """float64 normalisation, fp32 range, float64 affine map, cast to fp32."""
    if bits == 16:
        img = (planes[1].astype(np.uint16) << 8) + planes[0]
    else:
        img = planes[0] >> (8 - bits)
    norm = img / (2**bits - 1)  # float64
    rng = (np.asarray(maxs, np.float32) - np.asarray(mins, np.float32)).astype(np.float64)
    grid = norm.reshape(-1, len(np.atleast_1d(mins))) * rng + np.asarray(mins, np.float32).astype(np.float64)
    return grid.reshape(shape).astype(np.float32)
