"""On-disk quantized attribute format.

* ``grid_codec``: the min-max grid quantizer behind the reference's ``PngCompression``
  (gsplat/compression/png_compression.py) -- 8-bit, k-bit and 16-bit planes and their exact inverse -- as HIP kernels, plus
  the attribute-level pre/post-processing (log transform of the means, quaternion normalisation, square crop);
* ``decode``: the compressed planes decoded STRAIGHT INTO ``rasterization()``'s inputs by one kernel
  (``decode_to_rasterizer_inputs``), the K-means codebook of the higher SH bands (``kmeans_decode`` bit-exact against the
  reference's ``_decompress_kmeans``; ``kmeans_encode`` = seeded Lloyd iteration writing the same format), and the splat
  ordering in front of the grid codec (``sort_splats`` = PLAS, an external package as in the reference; ``morton_order`` =
  deterministic substitute).

* ``png_compression``: the file level -- ``PngCompression.compress(dir, splats)`` / ``decompress(dir)`` with the reference's
  directory layout (PNG image grids, ``shN.npz`` + ``mask.bin``, ``meta.json``) and a self-contained 8-bit PNG reader /
  writer (the image has no imageio), so directories written by either implementation are read by the other.
"""
from .decode import decode_to_rasterizer_inputs, kmeans_decode, kmeans_encode, morton_order, reorder_splats, sort_splats
from .png_compression import PngCompression, png_read, png_write
from .grid_codec import (
    compress_to_arrays,
    decompress_from_arrays,
    dequantize_grid,
    inverse_log_transform,
    log_transform,
    quantize_grid,
)

__all__ = ["quantize_grid", "dequantize_grid", "compress_to_arrays", "decompress_from_arrays", "log_transform",
           "inverse_log_transform", "decode_to_rasterizer_inputs", "kmeans_decode", "kmeans_encode", "morton_order",
           "sort_splats", "reorder_splats", "PngCompression", "png_read", "png_write"]
