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
