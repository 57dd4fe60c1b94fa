"""File level of the reference's PNG codec: ``PngCompression.compress(compress_dir, splats)`` /
``decompress(compress_dir)`` with the SAME directory layout (gsplat/compression/png_compression.py:17-162) --

    meta.json                      one entry per attribute (shape, dtype, mins, maxs, quantization, mask_bits, ...)
    means_l.png, means_u.png       low / high byte of the 16-bit grid of the log-transformed means
    scales.png quats.png sh0.png   k-bit grids, values in the top bits of the byte
    opacities.png                  8-bit grid
    shN.npz, mask.bin              K-means codebook (uint8 centroids, uint16 labels) of the splats that have higher SH
                                   bands at all, and the packed bit mask that says which
    <other>.npz                    anything else, under the key "arr"

so a directory written by either implementation is read by the other.  The quantisation arithmetic runs in the HIP kernels
of ``grid_codec`` / ``decode``; this module is the container around them:

* ``png_write`` / ``png_read``: 8-bit grey / grey+alpha / RGB / RGBA, non-interlaced -- what ``imageio.imwrite`` produces
  for these arrays (the image has no imageio / Pillow; a PNG is a zlib stream of filtered scanlines, PNG specification
  sections 5, 9, 10).  Reading undoes all five filter types (the sequential part is ``gs_png_unfilter`` of the library);
  writing picks None / Sub / Up per row by the usual minimum-sum-of-absolute-differences heuristic.
* the splat ordering is the one piece that needs an external package in the reference too (``plas``): ``use_sort=True``
  calls it and raises ImportError without it, ``use_sort="morton"`` uses the deterministic Morton order instead.
"""
from __future__ import annotations

import json
import os
import struct
import zlib
from dataclasses import dataclass
from typing import Any, Dict, Union

import numpy as np
import torch
from torch import Tensor

from .. import _backend as B
from .decode import kmeans_decode, kmeans_encode, morton_order, sort_splats
from .grid_codec import ATTRIBUTE_CODECS, _crop_to_square, dequantize_grid, inverse_log_transform, log_transform, quantize_grid

_PNG_MAGIC = b"\x89PNG\r\n\x1a\n"
_COLOR_TYPE = {1: 0, 2: 4, 3: 2, 4: 6}  # channels -> PNG colour type
_CHANNELS = {0: 1, 4: 2, 2: 3, 6: 4}


def _chunk(tag: bytes, data: bytes) -> bytes:
    return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xFFFFFFFF)


def png_write(path: str, img: np.ndarray, level: int = 6) -> None:
    """Write a uint8 image [H, W] or [H, W, C] (C = 1..4) as a PNG file."""
    img = np.ascontiguousarray(img)
    assert img.dtype == np.uint8 and img.ndim in (2, 3), (img.dtype, img.shape)
    if img.ndim == 2:
        img = img[:, :, None]
    h, w, c = img.shape
    assert c in _COLOR_TYPE and h > 0 and w > 0, img.shape
    rows = img.reshape(h, w * c)
    # per-row filter choice among None (0), Sub (1), Up (2): smallest sum of |signed residual|
    sub = rows.copy()
    sub[:, c:] = rows[:, c:] - rows[:, :-c]
    up = rows.copy()
    up[1:] = rows[1:] - rows[:-1]
    cand = np.stack([rows, sub, up])  # [3, h, w*c]
    cost = np.abs(cand.view(np.int8).astype(np.int32)).sum(axis=2)  # [3, h]
    ft = cost.argmin(axis=0).astype(np.uint8)  # [h]
    body = np.empty((h, 1 + w * c), dtype=np.uint8)
    body[:, 0] = ft
    body[:, 1:] = cand[ft, np.arange(h)]
    ihdr = struct.pack(">IIBBBBB", w, h, 8, _COLOR_TYPE[c], 0, 0, 0)
    with open(path, "wb") as f:
        f.write(_PNG_MAGIC + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(body.tobytes(), level)) + _chunk(b"IEND", b""))


def png_read(path: str) -> np.ndarray:
    """Read an 8-bit non-interlaced grey / grey+alpha / RGB / RGBA PNG -> uint8 [H, W] or [H, W, C] (as imageio.imread)."""
    with open(path, "rb") as f:
        buf = f.read()
    if buf[:8] != _PNG_MAGIC:
        raise ValueError(f"{path}: not a PNG file")
    pos, idat, hdr = 8, [], None
    while pos + 8 <= len(buf):
        (n,), tag = struct.unpack(">I", buf[pos:pos + 4]), buf[pos + 4:pos + 8]
        data = buf[pos + 8:pos + 8 + n]
        (crc,) = struct.unpack(">I", buf[pos + 8 + n:pos + 12 + n])
        if zlib.crc32(tag + data) & 0xFFFFFFFF != crc:
            raise ValueError(f"{path}: CRC mismatch in chunk {tag!r}")
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", data)
        elif tag == b"IDAT":
            idat.append(data)
        elif tag == b"IEND":
            break
        pos += 12 + n
    if hdr is None or not idat:
        raise ValueError(f"{path}: missing IHDR / IDAT")
    w, h, depth, ctype, comp, filt, interlace = hdr
    if depth != 8 or ctype not in _CHANNELS or comp != 0 or filt != 0 or interlace != 0:
        raise ValueError(f"{path}: only 8-bit non-interlaced grey / grey+alpha / RGB / RGBA images are supported "
                         f"(bit depth {depth}, colour type {ctype}, interlace {interlace})")
    c = _CHANNELS[ctype]
    raw = np.frombuffer(zlib.decompress(b"".join(idat)), dtype=np.uint8)
    if raw.size != h * (1 + w * c):
        raise ValueError(f"{path}: {raw.size} bytes of image data, expected {h * (1 + w * c)}")
    raw = np.ascontiguousarray(raw)
    out = np.empty((h, w * c), dtype=np.uint8)
    B.call("gs_png_unfilter", raw.ctypes.data, h, w * c, c, out.ctypes.data)
    return out.reshape(h, w) if c == 1 else out.reshape(h, w, c)


def _meta_of(params: Tensor) -> Dict[str, Any]:
    return {"shape": list(params.shape), "dtype": str(params.dtype).split(".")[1]}


@dataclass
class PngCompression:
    """The reference's ``PngCompression`` (same constructor arguments, same files).  ``use_sort``: True = PLAS (external
    package, as in the reference), "morton" = the deterministic Morton order, False = keep the order."""

    use_sort: Union[bool, str] = True
    verbose: bool = True
    n_clusters: int = 16384  # of the shN codebook (png_compression.py:132)
    opacity_threshold: float = 0.005  # outlier filter: sigmoid(opacity) below it is dropped (outlier_filter.py:8-9, 32-37)

    @torch.no_grad()
    def compress(self, compress_dir: str, splats: Dict[str, Tensor], entropy_models=None) -> None:
        if entropy_models is not None:
            raise ValueError("PngCompression should not require entropy_models")
        os.makedirs(compress_dir, exist_ok=True)
        splats = {k: v.detach() for k, v in splats.items()}
        keep = torch.sigmoid(splats["opacities"]) >= self.opacity_threshold
        splats = {k: v[keep] for k, v in splats.items()}
        splats["means"] = log_transform(splats["means"])
        splats["quats"] = torch.nn.functional.normalize(splats["quats"], dim=-1)
        n_before = len(splats["means"])
        splats, side = _crop_to_square(splats)
        if self.verbose and len(splats["means"]) != n_before:
            print(f"Warning: Number of Gaussians was not square. Removed {n_before - len(splats['means'])} Gaussians.")
        if self.use_sort == "morton":
            order = morton_order(splats["means"])
            splats = {k: v[order] for k, v in splats.items()}
        elif self.use_sort:
            splats = sort_splats(splats, verbose=self.verbose)

        meta: Dict[str, Any] = {}
        for name, value in splats.items():
            if value.numel() == 0:
                meta[name] = _meta_of(value)
            elif name in ATTRIBUTE_CODECS:
                bits, kind = ATTRIBUTE_CODECS[name]
                planes, meta[name] = quantize_grid(value, side, bits=bits, kbit=(kind == "kbit"))
                files = [f"{name}.png"] if bits != 16 else [f"{name}_l.png", f"{name}_u.png"]
                for fn, plane in zip(files, planes):
                    png_write(os.path.join(compress_dir, fn), plane.cpu().numpy())
            elif name == "shN":
                meta[name] = self._compress_masked_kmeans(compress_dir, value)
            else:
                np.savez_compressed(os.path.join(compress_dir, f"{name}.npz"), arr=value.cpu().numpy())
                meta[name] = _meta_of(value)
        with open(os.path.join(compress_dir, "meta.json"), "w") as f:
            json.dump(meta, f)

    def _compress_masked_kmeans(self, compress_dir: str, params: Tensor) -> Dict[str, Any]:
        """png_compression.py:521-600: the splats with any positive higher-band coefficient are clustered, the rest is a bit
        in mask.bin."""
        mask = (params > 0).any(dim=1).any(dim=1).reshape(-1)
        n = int(mask.numel())
        np.packbits(mask.cpu().numpy().astype(bool))[: (n + 7) // 8].tofile(os.path.join(compress_dir, "mask.bin"))
        if int(mask.sum()) == 0:  # nothing to cluster: an empty codebook, every splat decodes to zeros
            np.savez_compressed(os.path.join(compress_dir, "shN.npz"), centroids=np.zeros((0, params[0].numel()), np.uint8),
                                labels=np.zeros(0, np.uint16))
            meta = {**_meta_of(params), "mins": 0.0, "maxs": 0.0, "quantization": 8}
        else:
            cq, labels, meta = kmeans_encode(params[mask], n_clusters=self.n_clusters)
            np.savez_compressed(os.path.join(compress_dir, "shN.npz"), centroids=cq.cpu().numpy(),
                                labels=labels.cpu().numpy().astype(np.uint16))
        meta.update({"shape": list(params.shape), "mask_bits": n, "mask_byte": (n + 7) // 8})
        return meta

    @torch.no_grad()
    def decompress(self, compress_dir: str, device="cuda") -> Dict[str, Tensor]:
        with open(os.path.join(compress_dir, "meta.json"), "r") as f:
            meta = json.load(f)
        splats: Dict[str, Tensor] = {}
        for name, m in meta.items():
            if not np.all(m["shape"]):
                splats[name] = torch.zeros(m["shape"], dtype=getattr(torch, m["dtype"]), device=device)
            elif name in ATTRIBUTE_CODECS:
                bits, _ = ATTRIBUTE_CODECS[name]
                files = [f"{name}.png"] if bits != 16 else [f"{name}_l.png", f"{name}_u.png"]
                planes = [torch.from_numpy(png_read(os.path.join(compress_dir, fn))) for fn in files]
                splats[name] = dequantize_grid(planes, m, device=device)
            elif name == "shN":
                bits_loaded = np.fromfile(os.path.join(compress_dir, "mask.bin"), dtype=np.uint8)
                mask = torch.from_numpy(np.unpackbits(bits_loaded)[: m["mask_bits"]].astype(bool))
                z = np.load(os.path.join(compress_dir, "shN.npz"))
                if z["labels"].size == 0:
                    splats[name] = torch.zeros(m["shape"], dtype=getattr(torch, m["dtype"]), device=device)
                    continue
                splats[name] = kmeans_decode(torch.from_numpy(z["centroids"]), torch.from_numpy(z["labels"].astype(np.int32)), m,
                                             device=device, mask=mask)
            else:
                arr = np.load(os.path.join(compress_dir, f"{name}.npz"))["arr"]
                splats[name] = torch.tensor(arr).reshape(m["shape"]).to(dtype=getattr(torch, m["dtype"]), device=device)
        splats["means"] = inverse_log_transform(splats["means"])
        return splats
