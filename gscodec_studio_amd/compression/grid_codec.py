"""Min-max grid quantization of splat attributes (array level of the reference's PNG codec)."""
from __future__ import annotations

from typing import Any, Dict, List, Tuple

import torch
from torch import Tensor

from .. import _backend as B

# which attribute gets which quantizer (reference _get_compress_fn, png_compression.py:50-63):
# (bits, "plain" = _compress_png | "kbit" = _compress_png_kbit | "16" = _compress_png_16bit)
ATTRIBUTE_CODECS = {"means": (16, "16"), "scales": (8, "kbit"), "quats": (8, "kbit"), "opacities": (8, "plain"),
                    "sh0": (8, "kbit")}


def log_transform(x: Tensor) -> Tensor:
    """sign(x) log1p(|x|)  (gsplat/utils.py:36-37)."""
    return torch.sign(x) * torch.log1p(torch.abs(x))


def inverse_log_transform(y: Tensor) -> Tensor:
    """sign(y) expm1(|y|)  (gsplat/utils.py:40-41)."""
    return torch.sign(y) * torch.expm1(torch.abs(y))


def _stream(t: Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


@torch.no_grad()
def quantize_grid(params: Tensor, n_sidelen: int, bits: int = 8, kbit: bool = False) -> Tuple[List[Tensor], Dict[str, Any]]:
    """Quantize ``params`` ([n_sidelen^2, ...]) like _compress_png (bits=8), _compress_png_kbit (kbit=True, bits<=8)
    or _compress_png_16bit (bits=16).  Returns (planes, meta): uint8 images of shape [n_sidelen, n_sidelen, C]
    (squeezed like the reference's), one plane or (low, high) for 16 bits; meta has shape / dtype / mins / maxs
    (/ quantization for the k-bit variant), the reference's meta.json entry."""
    if not params.is_cuda:
        raise RuntimeError("quantize_grid: the HIP path needs device tensors (no CPU fallback)")
    assert bits == 16 or 1 <= bits <= 8, bits
    grid = params.reshape((n_sidelen, n_sidelen, -1)).contiguous().float()
    mins = torch.amin(grid, dim=(0, 1)).contiguous()
    maxs = torch.amax(grid, dim=(0, 1)).contiguous()
    C = grid.shape[-1]
    lo = torch.empty(grid.shape, dtype=torch.uint8, device=grid.device)
    hi = torch.empty(grid.shape, dtype=torch.uint8, device=grid.device) if bits == 16 else None
    with torch.cuda.device(grid.device):
        B.call("gs_grid_quantize", grid.numel(), C, B.ptr(grid), B.ptr(mins), B.ptr(maxs), bits, B.ptr(lo), B.ptr(hi), _stream(grid))
    meta = {"shape": list(params.shape), "dtype": str(params.dtype).split(".")[1], "mins": mins.tolist(), "maxs": maxs.tolist()}
    if kbit:
        meta["quantization"] = bits
    planes = [lo.squeeze()] if hi is None else [lo.squeeze(), hi.squeeze()]  # the reference squeezes single-channel images
    return planes, meta


@torch.no_grad()
def dequantize_grid(planes: List[Tensor], meta: Dict[str, Any], device=None) -> Tensor:
    """Inverse of ``quantize_grid`` (= _decompress_png / _decompress_png_kbit / _decompress_png_16bit), bit-exact."""
    shape = list(meta["shape"])
    dtype = getattr(torch, meta["dtype"])
    dev = torch.device(device) if device is not None else planes[0].device
    if dev.type != "cuda":
        raise RuntimeError("dequantize_grid: the HIP path needs device tensors (no CPU fallback)")
    bits = 16 if len(planes) == 2 else int(meta.get("quantization", 8))
    lo = planes[0].to(dev).contiguous()
    hi = planes[1].to(dev).contiguous() if len(planes) == 2 else None
    assert lo.dtype == torch.uint8 and (hi is None or hi.dtype == torch.uint8)
    mins = torch.tensor(meta["mins"], dtype=torch.float32, device=dev)
    maxs = torch.tensor(meta["maxs"], dtype=torch.float32, device=dev)
    C = mins.numel()
    out = torch.empty(lo.numel(), dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        B.call("gs_grid_dequantize", lo.numel(), C, B.ptr(lo), B.ptr(hi), B.ptr(mins), B.ptr(maxs), bits, B.ptr(out), _stream(lo))
    return out.reshape(shape).to(dtype)


def _crop_to_square(splats: Dict[str, Tensor]) -> Tuple[Dict[str, Tensor], int]:
    """Drop the lowest-opacity splats so that the count is a square (png_compression.py:112-119, 157-162)."""
    n = len(splats["means"])
    side = int(n**0.5)
    n_crop = n - side * side
    if n_crop:
        keep = torch.argsort(splats["opacities"], descending=True)[:-n_crop]
        splats = {k: v[keep] for k, v in splats.items()}
    return splats, side


@torch.no_grad()
def compress_to_arrays(splats: Dict[str, Tensor]) -> Tuple[Dict[str, List[Tensor]], Dict[str, Any]]:
    """Array-level ``PngCompression.compress`` for means / scales / quats / opacities / sh0 (pre-activation values, as the
    reference expects): log-transform the means, normalise the quaternions, crop to a square count, then quantize every
    attribute with its codec.  Other attributes (shN, ...) are passed through untouched under ``arrays[name] = [tensor]``
    with ``meta[name] = {"raw": True}`` (the reference uses a K-means codebook / npz for them).  No outlier filtering, no
    PLAS sort."""
    splats = dict(splats)
    splats["means"] = log_transform(splats["means"])
    splats["quats"] = torch.nn.functional.normalize(splats["quats"], dim=-1)
    splats, side = _crop_to_square(splats)
    arrays, meta = {}, {}
    for name, value in splats.items():
        if name in ATTRIBUTE_CODECS:
            bits, kind = ATTRIBUTE_CODECS[name]
            arrays[name], meta[name] = quantize_grid(value, side, bits=bits, kbit=(kind == "kbit"))
        else:
            arrays[name], meta[name] = [value], {"raw": True}
    return arrays, meta


@torch.no_grad()
def decompress_from_arrays(arrays: Dict[str, List[Tensor]], meta: Dict[str, Any], device="cuda") -> Dict[str, Tensor]:
    """Array-level ``PngCompression.decompress``: dequantize and undo the log transform of the means; the result is the
    splat dictionary ``rasterization()`` inputs are built from."""
    splats = {}
    for name, m in meta.items():
        splats[name] = arrays[name][0].to(device) if m.get("raw") else dequantize_grid(arrays[name], m, device=device)
    splats["means"] = inverse_log_transform(splats["means"])
    return splats
