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


def kmeans_decode(centroids_quant, labels, mins, maxs, bits, shape, mask=None):
    """_decompress_kmeans / _decompress_masked_kmeans (png_compression.py:487-520, 603-640): float64 normalisation of the
    codebook, fp32 scalar range, float64 affine map, gather by label, cast to fp32; masked-out splats are zero."""
    norm = centroids_quant / (2**bits - 1)  # float64
    rng = np.float64(np.float32(maxs) - np.float32(mins))
    cent = norm * rng + np.float64(np.float32(mins))
    rows = cent[np.asarray(labels).astype(np.int32)]
    if mask is not None:
        full = np.zeros((len(mask), rows.shape[1]), np.float64)
        full[np.asarray(mask, bool)] = rows
        rows = full
    return rows.reshape(shape).astype(np.float32)


def decode_pipeline(planes, mins, maxs, bits, shape, log_means=False):
    """One attribute of PngCompression.decompress: dequantize (+ the inverse log transform of the means, utils.py:40-41,
    in fp32 like torch: sign(y) * expm1(|y|))."""
    x = dequantize(planes, mins, maxs, bits, shape)
    if log_means:
        x = (np.sign(x) * np.expm1(np.abs(x))).astype(np.float32)
    return x
