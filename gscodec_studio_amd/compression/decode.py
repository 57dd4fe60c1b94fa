"""Decode of the compressed attribute planes straight into ``rasterization()``'s inputs, the K-means codebook of the higher
SH bands, and the splat ordering in front of the image-grid codec.

Reference: ``PngCompression.decompress`` (gsplat/compression/png_compression.py:166-236) + the activations of the eval path
(examples/simple_trainer.py:779-786).  The reference decodes on the HOST (numpy / torch CPU), uploads fp32 tensors and runs
exp / sigmoid / cat as separate kernels; ``decode_to_rasterizer_inputs`` uploads the uint8 planes (16 B per splat instead of
56 B) and one HIP kernel (csrc/codec.hip) writes every attribute ready for the renderer.  No torch / CPU fallback.
"""
from __future__ import annotations

import ctypes
from typing import Any, Dict, List, Optional, Tuple

import numpy as np
import torch
from torch import Tensor

from .. import _backend as B

_ORDER = ("means", "scales", "quats", "opacities", "sh0")
_WIDTH = {"means": 3, "scales": 3, "quats": 4, "opacities": 1, "sh0": 3}


def _stream(t: Tensor) -> int:
    return torch.cuda.current_stream(t.device).cuda_stream


def _bits(name: str, planes: List[Tensor], meta: Dict[str, Any]) -> int:
    return 16 if len(planes) == 2 else int(meta.get("quantization", 8))


@torch.no_grad()
def decode_to_rasterizer_inputs(arrays: Dict[str, List[Tensor]], meta: Dict[str, Any], device="cuda", activate: bool = True,
                                normalize_quats: bool = True) -> Dict[str, Tensor]:
    """``arrays`` / ``meta`` as produced by ``compress_to_arrays`` (= the reference's PNG planes and meta.json entries for
    means, scales, quats, opacities, sh0; anything else -- shN -- is passed through: raw tensors as they are, K-means entries
    through ``kmeans_decode``).  Returns ``{"means" [n,3], "quats" [n,4], "scales" [n,3], "opacities" [n], "sh0" [n,1,3],
    ...}`` on ``device``; with ``activate`` (default) scales = exp(.) and opacities = sigmoid(.), i.e. exactly the tensors
    ``rasterization(means, quats, scales, opacities, cat(sh0, shN), ...)`` takes."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("decode_to_rasterizer_inputs: the HIP path needs a GPU device (no CPU fallback)")
    for k in _ORDER:
        if k not in arrays:
            raise KeyError(f"decode_to_rasterizer_inputs: attribute {k!r} missing")
    n = int(np.prod(meta["means"]["shape"][:-1]))
    planes = {k: [p.to(dev).contiguous() for p in arrays[k]] for k in _ORDER}
    assert len(planes["means"]) == 2, "the means are stored as a 16-bit grid (two planes)"
    for k in _ORDER[1:]:
        assert len(planes[k]) == 1 and planes[k][0].dtype == torch.uint8 and planes[k][0].numel() == n * _WIDTH[k], k
    mins = (ctypes.c_float * 14)(*[float(np.float32(v)) for k in _ORDER for v in np.atleast_1d(meta[k]["mins"])])
    maxs = (ctypes.c_float * 14)(*[float(np.float32(v)) for k in _ORDER for v in np.atleast_1d(meta[k]["maxs"])])
    bits = (ctypes.c_uint32 * 5)(*[_bits(k, planes[k], meta[k]) for k in _ORDER])
    out = {"means": torch.empty((n, 3), device=dev), "scales": torch.empty((n, 3), device=dev), "quats": torch.empty((n, 4), device=dev),
           "opacities": torch.empty((n,), device=dev), "sh0": torch.empty((n, 1, 3), device=dev)}
    with torch.cuda.device(dev):
        B.call("gs_decode_splats", n, B.ptr(planes["means"][0]), B.ptr(planes["means"][1]), B.ptr(planes["scales"][0]),
               B.ptr(planes["quats"][0]), B.ptr(planes["opacities"][0]), B.ptr(planes["sh0"][0]), ctypes.addressof(mins),
               ctypes.addressof(maxs), ctypes.addressof(bits), int(normalize_quats), int(activate), B.ptr(out["means"]),
               B.ptr(out["scales"]), B.ptr(out["quats"]), B.ptr(out["opacities"]), B.ptr(out["sh0"]), _stream(out["means"]))
    for k, m in meta.items():
        if k in out:
            continue
        if m.get("raw"):
            out[k] = arrays[k][0].to(dev)
        elif len(arrays[k]) >= 2 and "quantization" in m:  # [centroids_quant, labels(, mask)] of the (masked) K-means codec
            out[k] = kmeans_decode(arrays[k][0], arrays[k][1], m, device=dev, mask=arrays[k][2] if len(arrays[k]) > 2 else None)
    return out


@torch.no_grad()
def kmeans_decode(centroids_quant: Tensor, labels: Tensor, meta: Dict[str, Any], device="cuda", mask: Optional[Tensor] = None) -> Tensor:
    """``_decompress_kmeans`` / ``_decompress_masked_kmeans`` (png_compression.py:487-520, 603-640) on the GPU, bit-exact:
    ``centroids_quant`` uint8 [k, w], ``labels`` integer [n] (or one per MASKED splat), meta with shape / mins / maxs /
    quantization; ``mask`` bool [n] (the unpacked mask.bin of the fork's masked variant): the other splats get zeros."""
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError("kmeans_decode: the HIP path needs a GPU device (no CPU fallback)")
    shape = list(meta["shape"])
    if not all(shape):
        return torch.zeros(shape, dtype=getattr(torch, meta["dtype"]), device=dev)
    cq = centroids_quant.to(dev).contiguous()
    lab = labels.to(dev).to(torch.int32).contiguous()
    assert cq.dtype == torch.uint8 and cq.dim() == 2
    out = torch.empty((lab.numel(), cq.shape[1]), dtype=torch.float32, device=dev)
    n_bad = torch.zeros(1, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        B.call("gs_kmeans_decode", lab.numel(), cq.shape[1], B.ptr(lab), B.ptr(cq), cq.shape[0], int(meta["quantization"]),
               float(np.float32(meta["mins"])), float(np.float32(meta["maxs"])), B.ptr(out), B.ptr(n_bad), _stream(out))
    # labels are file contents (shN.npz): the reference's `centroids[labels]` raises IndexError for one outside the
    # codebook; the kernel bounds-checks and counts them, and this read-back (a decode is not on the per-step path)
    # turns the count into the same error instead of a wild device read
    bad = int(n_bad.item())
    if bad:
        raise IndexError(f"kmeans_decode: {bad} label(s) outside the codebook of {cq.shape[0]} centroids "
                         "(truncated or mismatched shN.npz?)")
    if mask is not None:
        m = mask.to(dev).reshape(-1).to(torch.bool)
        rows = torch.nonzero_static(m, size=lab.numel()).view(-1)  # (no host synchronisation: the count is len(labels))
        full = torch.zeros((shape[0], cq.shape[1]), dtype=torch.float32, device=dev)
        full.index_copy_(0, rows, out)
        out = full
    return out.reshape(shape).to(getattr(torch, meta["dtype"]))


@torch.no_grad()
def kmeans_encode(params: Tensor, n_clusters: int = 65536, quantization: int = 8, iters: int = 10, seed: int = 0
                  ) -> Tuple[Tensor, Tensor, Dict[str, Any]]:
    """Codebook for the higher SH bands in the reference's on-disk form (``_compress_kmeans``, png_compression.py:420-484):
    returns (centroids_quant uint8 [k, w], labels int32 [n], meta).  The reference clusters with ``torchpq``'s K-means
    (manhattan distance, random initialisation) -- a package that is not in the image and whose result is not
    reproducible run to run; the clustering here is a plain Lloyd iteration in torch with the same distance and a seeded
    initialisation (device tensors, chunked assignment).  Everything AFTER the clustering -- the scalar min / max of the
    centroids, ``round((c - min) / (max - min) * (2^q - 1))`` in fp32, uint16-range labels -- is the reference's arithmetic,
    so ``kmeans_decode`` reads files written by either."""
    x = params.reshape(params.shape[0], -1).float()
    n, w = x.shape
    k = int(min(n_clusters, n))
    g = torch.Generator(device=x.device).manual_seed(seed)
    cent = x[torch.randperm(n, device=x.device, generator=g)[:k]].clone()
    labels = torch.zeros(n, dtype=torch.int64, device=x.device)
    chunk = max(1, (1 << 24) // max(k, 1))
    for _ in range(max(iters, 1)):
        for s in range(0, n, chunk):
            labels[s:s + chunk] = torch.cdist(x[s:s + chunk], cent, p=1).argmin(1)
        sums = torch.zeros_like(cent).index_add_(0, labels, x)
        cnts = torch.zeros(k, device=x.device).index_add_(0, labels, torch.ones(n, device=x.device))
        cent = torch.where(cnts[:, None] > 0, sums / cnts.clamp(min=1)[:, None], cent)
    mins, maxs = cent.min(), cent.max()
    norm = (cent - mins) / (maxs - mins)
    cq = (norm.cpu().numpy() * (2**quantization - 1)).round().astype(np.uint8)
    meta = {"shape": list(params.shape), "dtype": str(params.dtype).split(".")[1], "mins": mins.tolist(), "maxs": maxs.tolist(),
            "quantization": quantization}
    return torch.from_numpy(cq).to(params.device), labels.to(torch.int32), meta


def morton_order(means: Tensor, bits: int = 10) -> Tensor:
    """A deterministic spatially coherent ordering of the splats (30-bit Morton code of the means, ties by index): the
    stand-in for the PLAS sort when ``plas`` is not installed.  It only affects how well the image grids compress in a PNG
    container, never the decoded values."""
    lo, hi = means.amin(0), means.amax(0)
    q = ((means - lo) / (hi - lo).clamp(min=1e-12) * (2**bits - 1)).round().to(torch.int64).clamp(0, 2**bits - 1)
    code = torch.zeros(means.shape[0], dtype=torch.int64, device=means.device)
    for b in range(bits):
        for a in range(3):
            code |= ((q[:, a] >> b) & 1) << (3 * b + a)
    return torch.argsort(code, stable=True)


@torch.no_grad()
def reorder_splats(params, optimizers=None, perm: Optional[Tensor] = None, state: Optional[Dict[str, Tensor]] = None) -> Tensor:
    """Permute a trainer's splats IN MEMORY: every per-gaussian parameter, its optimizer state and the densification strategy's running
    statistics, along ``perm`` (default: ``morton_order(params["means"])``).  Rendering does not depend on the order of the splats, its
    speed does: the per-gaussian kernels read and write the rows of the gaussians a camera sees, and in a spatially sorted array those
    rows are neighbours (BASELINE config 2: -2 % per step against the fixture's order, -5 % against a shuffled array; config 5: -11..15 %;
    profiles/r06_splat_order.txt).  Densification appends new splats at the end, so call it after the set changed (or every few
    thousand steps): one gather per tensor.

    ``params``: dict / ParameterDict name -> Parameter [N, ...]; ``optimizers``: dict name -> Optimizer holding that parameter (the
    layout of the reference's trainers and of its ``strategy/ops.py:_update_param_with_optimizer``, whose replace-the-parameter-and-move-
    the-state mechanics this follows); ``state``: the strategy's per-gaussian tensors (``grad2d``, ``count``, ``radii`` ...), permuted in
    place of the dict.  Tensors whose first dimension is not N (an MLP decoder's weights, scalar steps) are left alone.  Returns perm."""
    n = int(params["means"].shape[0])
    if perm is None:
        perm = morton_order(params["means"].detach())
    assert perm.shape == (n,), perm.shape
    for name in list(params.keys()):
        p = params[name]
        if p.dim() == 0 or p.shape[0] != n:
            continue
        new = torch.nn.Parameter(p.detach()[perm].contiguous(), requires_grad=p.requires_grad)
        opt = None if optimizers is None else optimizers.get(name)
        if opt is not None:
            for group in opt.param_groups:
                for i, q in enumerate(group["params"]):
                    if q is p:
                        group["params"][i] = new
            st = opt.state.pop(p, None)
            if st is not None:
                opt.state[new] = {k: (v[perm].contiguous() if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n else v)
                                  for k, v in st.items()}
        params[name] = new
    if state is not None:
        for k, v in list(state.items()):
            if torch.is_tensor(v) and v.dim() > 0 and v.shape[0] == n:
                state[k] = v[perm].contiguous()
    return perm


def sort_splats(splats: Dict[str, Tensor], verbose: bool = True, return_indices: bool = False, sort_with_shN: bool = False):
    """The reference's ``sort_splats`` (gsplat/compression/sort.py:7-59): Parallel Linear Assignment Sorting of the splats
    on the square grid, through the external ``plas`` package -- a randomised heuristic (``torch.randperm`` start) whose
    output order changes the PNG size only.  ``plas`` is a hard dependency of that ordering and is not in the image: like
    the reference this raises ``ImportError`` when it is missing; ``morton_order`` is the deterministic substitute."""
    try:
        from plas import sort_with_plas
    except Exception as e:  # noqa: BLE001
        raise ImportError("Please install PLAS with 'pip install git+https://github.com/fraunhoferhhi/PLAS.git' to use sorting "
                          "(or order the splats with gscodec_studio_amd.compression.morton_order)") from e
    n_gs = len(splats["means"])
    n_sidelen = int(n_gs**0.5)
    assert n_sidelen**2 == n_gs, "Must be a perfect square"
    keys = [k for k in splats if sort_with_shN or k != "shN"]
    params = torch.cat([splats[k].reshape(n_gs, -1) for k in keys], dim=-1)
    shuffled = torch.randperm(n_gs, device=params.device)
    grid = params[shuffled].reshape((n_sidelen, n_sidelen, -1))
    _, idx = sort_with_plas(grid.permute(2, 0, 1), improvement_break=1e-4, verbose=verbose)
    idx = shuffled[idx.squeeze().flatten()]
    out = {k: v[idx] for k, v in splats.items()}
    return (out, idx) if return_indices else out
