"""On-disk quantized attribute format, array level.

The min-max grid quantizer behind the reference's ``PngCompression`` (gsplat/compression/png_compression.py)
-- 8-bit, k-bit and 16-bit planes and their exact inverse -- as HIP kernels, plus the attribute-level
pre/post-processing (log transform of the means, quaternion normalisation, square crop).  The lossless
containers (PNG via ``imageio``), the PLAS sort and the K-means codebook for shN are NOT built: none of those
packages is in the image, and they are CPU / library code outside the GPU hot path.
"""
from .grid_codec import (
    compress_to_arrays,
    decompress_from_arrays,
    dequantize_grid,
    inverse_log_transform,
    log_transform,
    quantize_grid,
)

__all__ = ["quantize_grid", "dequantize_grid", "compress_to_arrays", "decompress_from_arrays", "log_transform",
           "inverse_log_transform"]
