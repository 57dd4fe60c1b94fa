"""gscodec_studio_amd -- MI355X-native rasterize + quantize hot path of GSCodec Studio.

Public surface (mirrors the reference's ``gsplat`` package for this path only):
  rendering.rasterization, the operator functions of ``_wrapper`` and
  ``compression_simulation.{CompressionSimulation, STGCompressionSimulation, fake_quantize_ste, STE}``.
"""
from ._wrapper import (
    accumulate,
    fully_fused_projection,
    isect_offset_encode,
    isect_tiles,
    persp_proj,
    proj,
    quat_scale_to_covar_preci,
    rasterize_to_indices_in_range,
    rasterize_to_pixels,
    spherical_harmonics,
    spherical_harmonics_shared,
    world_to_cam,
)
from .rendering import rasterization
from .version import __version__


def __getattr__(name):  # (lazy: the codec module is not on the training path)
    if name == "PngCompression":  # reference: gsplat/__init__.py:3
        from .compression import PngCompression

        return PngCompression
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


__all__ = [
    "rasterization", "fully_fused_projection", "spherical_harmonics", "spherical_harmonics_shared",
    "isect_tiles", "isect_offset_encode", "rasterize_to_pixels", "quat_scale_to_covar_preci", "proj", "persp_proj",
    "world_to_cam", "rasterize_to_indices_in_range", "accumulate", "PngCompression", "__version__",
]
