"""CPU: the oracle (oracle/gs_oracle.c) against the golden vectors generated from the
reference's own Python functions (tests/golden/make_golden.py).  This re-checks, without the
reference present, what make_golden.py asserted when it pinned the oracle."""
import numpy as np
import pytest

from oracle import gs_oracle as O
from util import assert_close, golden, rel_l2


@pytest.mark.parametrize("model", ["pinhole", "ortho", "fisheye"])
@pytest.mark.parametrize("comp", [False, True])
def test_projection(model, comp):
    gd = golden("projection.npz")
    tag = f"{model}_{int(comp)}"
    W, H = int(gd["width"]), int(gd["height"])
    r, m2, d, c, cp = O.projection_fwd(gd["means"], None, gd["quats"], gd["scales"], gd["viewmats"], gd["Ks"], W, H,
                                       calc_compensations=comp, camera_model=model)
    ref_r = gd[f"{tag}_radii"]
    assert np.abs(r - ref_r).max() <= 1 and (r == ref_r).mean() > 0.999
    v = (r > 0) & (ref_r > 0)
    assert_close(m2[v], gd[f"{tag}_means2d"][v], 1e-5, 2e-4, "means2d")
    assert_close(d[v], gd[f"{tag}_depths"][v], 1e-6, 1e-6, "depths")
    assert_close(c[v], gd[f"{tag}_conics"][v], 3e-4, 1e-5, "conics")
    if comp:
        assert_close(cp[v], gd[f"{tag}_comp"][v], 1e-4, 1e-3, "compensations")
    valid = ref_r > 0
    vm = gd["v_means2d"] * valid[..., None]
    vd = gd["v_depths"] * valid
    vc = gd["v_conics"] * valid[..., None]
    vcp = gd["v_comp"] * valid if comp else None
    b_m, _, b_q, b_s, b_v = O.projection_bwd(gd["means"], None, gd["quats"], gd["scales"], gd["viewmats"], gd["Ks"], W, H,
                                             0.3, model, ref_r, gd[f"{tag}_conics"], gd[f"{tag}_comp"] if comp else None,
                                             vm, vd, vc, vcp)
    for name, got in (("v_means", b_m), ("v_quats", b_q), ("v_scales", b_s), ("v_viewmats", b_v)):
        assert rel_l2(got, gd[f"{tag}_{name}"]) < 2e-3, name


@pytest.mark.parametrize("deg", [0, 1, 2, 3, 4])
def test_sh(deg):
    gd = golden("sh.npz")
    assert_close(O.sh_fwd(deg, gd["dirs"], gd["coeffs"]), gd[f"deg{deg}_colors"], 1e-5, 1e-5, "colors")
    vc, vd = O.sh_bwd(deg, gd["dirs"], gd["coeffs"], gd["v_colors"])
    assert_close(vc, gd[f"deg{deg}_v_coeffs"], 1e-5, 1e-5, "v_coeffs")
    assert_close(vd, gd[f"deg{deg}_v_dirs"], 1e-4, 1e-4, "v_dirs")


@pytest.mark.parametrize("case", ["a", "b"])
def test_isect_bit_exact(case):
    gd = golden("isect.npz")
    ts, tw, th = int(gd[f"{case}_tile_size"]), int(gd[f"{case}_tile_width"]), int(gd[f"{case}_tile_height"])
    tpg, ids, flat = O.isect_tiles(gd[f"{case}_means2d"], gd[f"{case}_radii"], gd[f"{case}_depths"], ts, tw, th)
    assert np.array_equal(tpg, gd[f"{case}_tiles_per_gauss"])
    assert np.array_equal(ids, gd[f"{case}_isect_ids"])
    assert np.array_equal(flat, gd[f"{case}_flatten_ids"])
    C = gd[f"{case}_radii"].shape[0]
    assert np.array_equal(O.isect_offset_encode(ids, C, tw, th), gd[f"{case}_isect_offsets"])


def test_sort_is_stable_and_signed_at_64_bits():
    rs = np.random.RandomState(0)
    keys = (rs.randint(0, 50, size=20000).astype(np.int64) << 33) | (rs.randint(0, 2, size=20000).astype(np.int64) << 63)
    vals = np.arange(20000, dtype=np.int32)
    k, v = O.sort_pairs(keys, vals, 64)
    order = np.argsort(keys, kind="stable")
    assert np.array_equal(v, vals[order]) and np.array_equal(k, keys[order])
    k, v = O.sort_pairs(keys, vals, 40)
    order = np.argsort(keys & ((1 << 40) - 1), kind="stable")
    assert np.array_equal(v, vals[order])


@pytest.mark.parametrize("name", ["scales", "quats", "opacities", "sh0", "stg_opacities", "stg_colors", "features"])
def test_quantizers_bit_exact(name):
    gd = golden("quantize.npz")
    lo, hi = (float(x) for x in gd[f"{name}_bounds"])
    x = gd[f"{name}_x"]
    for bits in (8, 4):
        xa, out = O.quant_round_fwd(x, lo, hi, bits)
        assert np.array_equal(out.view(np.uint32), gd[f"{name}_round{bits}_out"].view(np.uint32))
        assert np.array_equal(xa.view(np.uint32), gd[f"{name}_round{bits}_x_after"].view(np.uint32))
        q = float(gd[f"{name}_noise{bits}_q_step"])
        o = O.quant_noise_fwd(x, gd[f"{name}_noise{bits}_noise"], lo, hi, q)
        assert np.array_equal(o.view(np.uint32), gd[f"{name}_noise{bits}_out"].view(np.uint32))
        g = O.quant_noise_bwd(x, gd[f"{name}_noise{bits}_v_out"], lo, hi)
        assert np.array_equal(g.view(np.uint32), gd[f"{name}_noise{bits}_v_x"].view(np.uint32))


def test_quantizer_edge_vector():
    """6-value edge vector, bounds +-1, 8 bit."""
    xa, out = O.quant_round_fwd(np.array([-3, -0.4, 0, 0.30001, 0.9, 2.5], np.float32), -1, 1, 8)
    assert_close(out, [-1, -0.4039216, -0.0039216, 0.3019608, 0.8980392, 1], 0, 1e-6, "edge vector")
    assert_close(xa, [-1, -0.4, 0, 0.30001, 0.9, 1], 0, 1e-7, "in-place clamp")
