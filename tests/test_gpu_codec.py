"""GPU: gs_grid_quantize / gs_grid_dequantize BIT-EXACT against the golden vectors recorded from the reference's
PNG codec functions, and an encode -> decode round trip of a full splat dictionary at BASELINE size."""
import numpy as np
import pytest
import torch

from util import N, T, golden

pytestmark = pytest.mark.gpu

from test_codec_cpu import CASES  # noqa: E402


@pytest.mark.parametrize("name", CASES)
def test_grid_quantizer_bit_exact(name):
    from gscodec_studio_amd.compression import dequantize_grid, quantize_grid

    gd = golden("codec.npz")
    side, bits = int(gd["n_sidelen"]), int(gd[f"{name}.bits"])
    x = T(gd[f"{name}.x"])
    planes, meta = quantize_grid(x, side, bits=bits, kbit=name.endswith("k"))
    assert np.array_equal(np.asarray(meta["mins"], np.float32), gd[f"{name}.mins"])
    assert np.array_equal(np.asarray(meta["maxs"], np.float32), gd[f"{name}.maxs"])
    for i, p in enumerate(planes):
        assert p.dtype == torch.uint8 and np.array_equal(N(p), gd[f"{name}.plane{i}"])
    dec = dequantize_grid([T(gd[f"{name}.plane{i}"]) for i in range(len(planes))], meta)
    assert dec.shape == x.shape and dec.dtype == torch.float32
    assert np.array_equal(N(dec).view(np.uint32), gd[f"{name}.decoded"].view(np.uint32))


def test_splat_dictionary_round_trip_full_size():
    from gscodec_studio_amd.compression import compress_to_arrays, decompress_from_arrays

    g = torch.Generator(device="cuda").manual_seed(0)
    n = 1_006_065  # not a square: 1003^2 = 1,006,009 -> 56 lowest-opacity splats are dropped
    splats = {
        "means": torch.randn(n, 3, device="cuda", generator=g) * 3,
        "scales": torch.rand(n, 3, device="cuda", generator=g) * 8 - 9,
        "quats": torch.randn(n, 4, device="cuda", generator=g),
        "opacities": torch.randn(n, device="cuda", generator=g) * 3,
        "sh0": torch.randn(n, 1, 3, device="cuda", generator=g),
        "shN": torch.randn(n, 15, 3, device="cuda", generator=g) * 0.05,
    }
    arrays, meta = compress_to_arrays(splats)
    side = 1003
    assert arrays["means"][0].shape == (side, side, 3) and len(arrays["means"]) == 2 and arrays["opacities"][0].shape == (side, side)
    assert meta["scales"]["quantization"] == 8 and "quantization" not in meta["opacities"] and meta["shN"] == {"raw": True}
    out = decompress_from_arrays(arrays, meta)
    keep = torch.argsort(splats["opacities"], descending=True)[: side * side]
    # quantization error bounds: half a step of the per-channel range (means: in log space)
    def check(name, ref, bits):
        got = out[name]
        rng = (ref.reshape(side * side, -1).amax(0) - ref.reshape(side * side, -1).amin(0))
        err = (got - ref).abs().reshape(side * side, -1).amax(0)
        assert bool((err <= 0.5 * rng / (2**bits - 1) * (1 + 1e-3) + 1e-6).all()), (name, err, rng)
    check("scales", splats["scales"][keep], 8)
    check("opacities", splats["opacities"][keep], 8)
    check("sh0", splats["sh0"][keep], 8)
    check("quats", torch.nn.functional.normalize(splats["quats"][keep], dim=-1), 8)
    from gscodec_studio_amd.compression import log_transform
    lm = log_transform(splats["means"][keep])
    lerr = (log_transform(out["means"]) - lm).abs().amax(0)
    assert bool((lerr <= 0.5 * (lm.amax(0) - lm.amin(0)) / 65535 * 1.01 + 1e-5).all())
    assert torch.equal(out["shN"], splats["shN"][keep])
    # idempotence: re-encoding the decoded attributes reproduces the same planes (scales: min/max are grid points)
    arrays2, _ = compress_to_arrays({k: v for k, v in out.items()})
    assert torch.equal(arrays2["scales"][0], arrays["scales"][0]) and torch.equal(arrays2["opacities"][0], arrays["opacities"][0])


def _pipeline_arrays():
    gd = golden("codec_pipeline.npz")
    arrays, meta = {}, {}
    for name in ("means", "scales", "quats", "opacities", "sh0"):
        bits = int(gd[f"{name}.bits"])
        arrays[name] = [T(gd[f"{name}.plane0"])] + ([T(gd[f"{name}.plane1"])] if bits == 16 else [])
        meta[name] = {"shape": [int(v) for v in gd[f"{name}.shape"]], "dtype": "float32", "mins": gd[f"{name}.mins"].tolist(),
                      "maxs": gd[f"{name}.maxs"].tolist()}
        if bits not in (8, 16) or name in ("scales", "quats", "sh0"):
            meta[name]["quantization"] = bits
    arrays["shN"] = [T(gd["shN.centroids"]), T(gd["shN.labels"].astype(np.int32)), T(gd["shN.mask"])]
    meta["shN"] = {"shape": list(gd["shN.decoded"].shape), "dtype": "float32", "mins": float(gd["shN.mins"]), "maxs": float(gd["shN.maxs"]),
                   "quantization": 8}
    return gd, arrays, meta


def test_fused_decode_matches_reference_decompress():
    """decode_to_rasterizer_inputs on the planes of a directory the REFERENCE compressed and decompressed
    (tests/golden/make_golden_codec_pipeline.py): raw parameters bit for bit (means: within 2 ulp, expm1f vs torch.expm1),
    the masked K-means shN bit for bit; with activations on, what exp / sigmoid of the reference's output give."""
    from gscodec_studio_amd.compression import decode_to_rasterizer_inputs

    gd, arrays, meta = _pipeline_arrays()
    raw = decode_to_rasterizer_inputs(arrays, meta, activate=False, normalize_quats=False)
    for name in ("scales", "quats", "opacities", "sh0", "shN"):
        assert raw[name].dtype == torch.float32 and tuple(raw[name].shape) == gd[f"{name}.decoded"].shape, name
        assert np.array_equal(N(raw[name]).view(np.uint32), gd[f"{name}.decoded"].view(np.uint32)), name
    ref_m = gd["means.decoded"]
    assert np.all(np.abs(N(raw["means"]) - ref_m) <= 2.4e-7 * np.abs(ref_m) + 1e-30)
    act = decode_to_rasterizer_inputs(arrays, meta)  # what rasterization() takes
    assert np.allclose(N(act["scales"]), np.exp(gd["scales.decoded"]), rtol=2e-6, atol=0)
    assert np.allclose(N(act["opacities"]), 1 / (1 + np.exp(-gd["opacities.decoded"].astype(np.float64))), rtol=2e-6, atol=1e-9)
    q = gd["quats.decoded"]
    assert np.allclose(N(act["quats"]), q / np.linalg.norm(q, axis=-1, keepdims=True), rtol=2e-6, atol=1e-7)
    assert torch.equal(act["sh0"], raw["sh0"]) and torch.equal(act["means"], raw["means"])


def test_decode_then_render_full_size_and_kmeans_round_trip():
    """1,006,009 splats (1003^2): compress_to_arrays -> decode_to_rasterizer_inputs == decompress_from_arrays + activations,
    the decoded splats render, and a K-means codebook written by kmeans_encode is read back by kmeans_decode."""
    from gscodec_studio_amd import rasterization
    from gscodec_studio_amd._helper import sh_workload
    from gscodec_studio_amd.compression import (compress_to_arrays, decode_to_rasterizer_inputs, decompress_from_arrays, kmeans_decode,
                                                kmeans_encode, morton_order)

    w = sh_workload(scene_grid=3, device="cuda:0")
    side = 1003
    order = morton_order(w["means"])
    assert torch.equal(torch.sort(order).values, torch.arange(w["N"], device="cuda:0"))  # a permutation
    splats = {"means": w["means"], "scales": w["scales"].clamp_min(1e-6).log(), "quats": w["quats"],
              "opacities": torch.logit(w["opacities"].clamp(1e-4, 1 - 1e-4)), "sh0": w["sh"][:, :1].contiguous()}
    splats = {k: v[order] for k, v in splats.items()}
    arrays, meta = compress_to_arrays(splats)
    dec = decode_to_rasterizer_inputs(arrays, meta)
    ref = decompress_from_arrays(arrays, meta)
    assert dec["means"].shape == (side * side, 3)
    assert torch.allclose(dec["means"], ref["means"], rtol=3e-7, atol=0)
    assert torch.allclose(dec["scales"], torch.exp(ref["scales"]), rtol=2e-6) and torch.allclose(dec["opacities"], torch.sigmoid(ref["opacities"]), rtol=2e-6, atol=1e-9)
    assert torch.equal(dec["sh0"], ref["sh0"])
    vm, Ks = w["viewmats"][:1], w["Ks"][:1]
    rc, ra, _ = rasterization(dec["means"], dec["quats"], dec["scales"], dec["opacities"], dec["sh0"], vm, Ks, w["width"], w["height"],
                              sh_degree=0, packed=False)
    assert bool(torch.isfinite(rc).all()) and float(ra.max()) > 0.5
    # K-means: encode 50 k rows of higher bands into 256 centroids, decode, error bounded by cluster radius + quantization
    shn = w["sh"][:50_000, 1:].contiguous()
    cq, labels, m = kmeans_encode(shn, n_clusters=256, iters=4)
    assert cq.dtype == torch.uint8 and cq.shape == (256, 45) and labels.dtype == torch.int32 and int(labels.max()) < 256
    back = kmeans_decode(cq, labels, m)
    assert back.shape == shn.shape
    step = (m["maxs"] - m["mins"]) / 255
    cent = back.reshape(50_000, -1)
    # every row decodes to ITS centroid: rows with equal labels decode identically, and the centroid is the cluster mean up to q/2
    l0 = int(labels[0])
    same = (labels == l0)
    assert bool((cent[same] == cent[same][0]).all())
    mean0 = shn.reshape(50_000, -1)[same].mean(0)
    assert float((cent[same][0] - mean0).abs().max()) <= 0.5 * step * 1.01 + 1e-6
    # labels are file contents: one outside the codebook raises (as the reference's centroids[labels] does), no wild read
    bad = labels.clone()
    bad[17] = 256
    with pytest.raises(IndexError):
        kmeans_decode(cq, bad, m)
    bad[17] = -1
    with pytest.raises(IndexError):
        kmeans_decode(cq, bad, m)


def test_png_directory_written_like_the_reference_is_decoded_bit_exact(tmp_path):
    """File level: the golden planes laid out on disk the way PngCompression.compress of the reference does (image grids as
    PNG files -- written with Sub / Up / Average / Paeth rows like imageio's encoder chooses them --, shN.npz, mask.bin,
    meta.json) and read back by ``PngCompression.decompress``: what the reference's own decompress returned for the same
    planes (tests/golden/make_golden_codec_pipeline.py)."""
    import json
    import struct
    import zlib

    from test_png_cpu import CTYPE, MAGIC, _chunk, _filter_rows

    from gscodec_studio_amd.compression import PngCompression

    gd, arrays, meta = _pipeline_arrays()
    side = int(gd["n_sidelen"])

    def write_png(name, plane):
        img = plane.reshape(side, side, -1)
        types = [(y * 7 + 3) % 5 for y in range(side)]
        ihdr = struct.pack(">IIBBBBB", side, side, 8, CTYPE[img.shape[2]], 0, 0, 0)
        (tmp_path / name).write_bytes(MAGIC + _chunk(b"IHDR", ihdr) + _chunk(b"IDAT", zlib.compress(_filter_rows(img, types))) + _chunk(b"IEND", b""))

    for name in ("means", "scales", "quats", "opacities", "sh0"):
        if int(gd[f"{name}.bits"]) == 16:
            write_png(f"{name}_l.png", gd[f"{name}.plane0"])
            write_png(f"{name}_u.png", gd[f"{name}.plane1"])
        else:
            write_png(f"{name}.png", gd[f"{name}.plane0"])
    mask = gd["shN.mask"].astype(bool)
    np.packbits(mask)[: (len(mask) + 7) // 8].tofile(str(tmp_path / "mask.bin"))
    np.savez_compressed(str(tmp_path / "shN.npz"), centroids=gd["shN.centroids"], labels=gd["shN.labels"].astype(np.uint16))
    meta["shN"].update({"mask_bits": len(mask), "mask_byte": (len(mask) + 7) // 8})
    (tmp_path / "meta.json").write_text(json.dumps(meta))

    out = PngCompression().decompress(str(tmp_path))
    for name in ("scales", "quats", "opacities", "sh0", "shN"):
        assert np.array_equal(N(out[name]).view(np.uint32), gd[f"{name}.decoded"].view(np.uint32)), name
    ref_m = gd["means.decoded"]
    assert np.all(np.abs(N(out["means"]) - ref_m) <= 2.4e-7 * np.abs(ref_m) + 1e-30)


def test_png_compression_directory_round_trip(tmp_path):
    """compress -> files -> decompress equals the array-level pipeline on the same (filtered, cropped, ordered) splats;
    PLAS ordering needs the external package exactly as in the reference."""
    import os

    from gscodec_studio_amd.compression import PngCompression, compress_to_arrays, decompress_from_arrays, morton_order

    g = torch.Generator(device="cpu").manual_seed(5)
    n = 70 * 70 + 37
    splats = {"means": torch.randn(n, 3, generator=g) * 4, "scales": torch.randn(n, 3, generator=g) - 3,
              "quats": torch.randn(n, 4, generator=g), "opacities": torch.randn(n, generator=g) * 3,
              "sh0": torch.randn(n, 1, 3, generator=g), "shN": torch.randn(n, 15, 3, generator=g) * 0.1,
              "features": torch.randn(n, 5, generator=g)}
    splats["shN"][::3] = -splats["shN"][::3].abs()  # a third of the splats has no positive higher-band coefficient: masked out
    splats = {k: v.cuda() for k, v in splats.items()}
    with pytest.raises(ImportError):
        PngCompression(verbose=False).compress(str(tmp_path / "plas"), dict(splats))

    for mode in (False, "morton"):
        d = str(tmp_path / f"dir_{mode}")
        PngCompression(use_sort=mode, verbose=False, n_clusters=256).compress(d, dict(splats))
        assert sorted(os.listdir(d)) == sorted(["meta.json", "means_l.png", "means_u.png", "scales.png", "quats.png", "opacities.png",
                                                "sh0.png", "shN.npz", "mask.bin", "features.npz"])
        out = PngCompression().decompress(d)
        # the same splats through the array-level pipeline
        keep = torch.sigmoid(splats["opacities"]) >= 0.005
        kept = {k: v[keep] for k, v in splats.items()}
        side = int(len(kept["means"]) ** 0.5)
        crop = torch.argsort(kept["opacities"], descending=True)[: side * side]
        kept = {k: v[crop] for k, v in kept.items()}
        if mode == "morton":
            from gscodec_studio_amd.compression import log_transform
            order = morton_order(log_transform(kept["means"]))
            kept = {k: v[order] for k, v in kept.items()}
        arrays, meta = compress_to_arrays({k: kept[k] for k in ("means", "scales", "quats", "opacities", "sh0")})
        want = decompress_from_arrays(arrays, meta)
        for k in ("means", "scales", "quats", "opacities", "sh0"):
            assert torch.equal(out[k], want[k]), (mode, k)
        assert torch.equal(out["features"], kept["features"])
        has = (kept["shN"] > 0).any(dim=1).any(dim=1)
        assert bool((out["shN"][~has] == 0).all()) and out["shN"].shape == kept["shN"].shape
        # a 256-entry codebook of 15x3 coefficients: coarse, but every decoded row must be one codebook row
        rows = out["shN"][has].reshape(int(has.sum()), -1)
        assert torch.unique(rows, dim=0).shape[0] <= 256
        assert float((rows - kept["shN"][has].reshape(rows.shape)).abs().mean()) < 0.2
