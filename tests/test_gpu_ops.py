"""GPU parity tests of the individual HIP stages against the CPU oracle and the golden
vectors produced by the reference (tests/golden/make_golden.py).  Test recipes follow the
reference's tests/test_basic.py (cited per test).  All calls go through the C ABI.

Bars (BASELINE.json north_star): tile/bin indices bit-exact; fp32 results within 1e-4
relative (plus an absolute floor for values near zero), tolerances written per assert.
"""
import math

import numpy as np
import pytest
import torch

from util import N, T, assert_close, garden, golden, rel_l2

pytestmark = pytest.mark.gpu

from oracle import gs_oracle as O  # noqa: E402


@pytest.fixture(scope="module")
def ops():
    import gscodec_studio_amd as g

    return g


# ---------------------------------------------------------------------------
# projection  (reference tests/test_basic.py:176-279)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("camera_model", ["pinhole", "ortho", "fisheye"])
@pytest.mark.parametrize("calc_compensations", [False, True])
def test_projection_vs_golden_and_oracle(ops, camera_model, calc_compensations):
    gd = golden("projection.npz")
    tag = f"{camera_model}_{int(calc_compensations)}"
    W, H = int(gd["width"]), int(gd["height"])
    means, quats, scales = T(gd["means"], True), T(gd["quats"], True), T(gd["scales"], True)
    viewmats, Ks = T(gd["viewmats"], True), T(gd["Ks"])
    radii, means2d, depths, conics, comps = ops.fully_fused_projection(
        means, None, quats, scales, viewmats, Ks, W, H, calc_compensations=calc_compensations,
        camera_model=camera_model)
    r_ref = gd[f"{tag}_radii"]
    r = N(radii)
    # reference test allows +-1 on radii (fast-math); against the reference's torch path we are exact here
    assert np.abs(r - r_ref).max() <= 1
    assert (r == r_ref).mean() > 0.999
    valid = (r > 0) & (r_ref > 0)
    assert_close(N(means2d)[valid], gd[f"{tag}_means2d"][valid], 1e-4, 1e-4, "means2d")
    assert_close(N(depths)[valid], gd[f"{tag}_depths"][valid], 1e-4, 1e-5, "depths")
    assert_close(N(conics)[valid], gd[f"{tag}_conics"][valid], 3e-4, 1e-5, "conics")
    if calc_compensations:
        assert_close(N(comps)[valid], gd[f"{tag}_comp"][valid], 1e-4, 1e-3, "compensations")
    else:
        assert comps is None

    # backward with the golden cotangents, masked by the reference's validity
    vmask = torch.as_tensor(r_ref > 0, device=means.device)
    loss = (means2d * T(gd["v_means2d"]) * vmask[..., None]).sum() + (depths * T(gd["v_depths"]) * vmask).sum() + \
           (conics * T(gd["v_conics"]) * vmask[..., None]).sum()
    if calc_compensations:
        loss = loss + (comps * T(gd["v_comp"]) * vmask).sum()
    g_m, g_q, g_s, g_v = torch.autograd.grad(loss, (means, quats, scales, viewmats))
    # Ground truth: the REFERENCE's projection evaluated in float64 on the same inputs / cotangents / visibility
    # (tests/golden/make_golden_projection_f64.py).  Measured on MI355X (round 4): HIP 6e-8 ... 3.7e-5 relative L2 from it, the
    # reference's own fp32 run (the golden vectors) 6e-8 ... 1.05e-4 (fisheye v_scales).  Asserted: the north-star 1e-4 against
    # float64; against the fp32 golden vectors 1e-4 + their own distance from float64; elementwise (small entries are
    # differences of large terms in both implementations: the reference's fp32 run is off by up to 1.4 % of |entry| + 1e-4
    # max|entry| there) no worse than 3x the reference's own worst entry.
    g64 = golden("projection_f64.npz")
    for name, got in (("v_means", g_m), ("v_quats", g_q), ("v_scales", g_s), ("v_viewmats", g_v)):
        want, ref32 = g64[f"{tag}_{name}"], gd[f"{tag}_{name}"]
        e_hip, e_ref = rel_l2(N(got), want), rel_l2(ref32, want)
        assert e_hip < 1e-4, (name, e_hip)
        assert rel_l2(N(got), ref32) < 1e-4 + e_ref, (name, rel_l2(N(got), ref32), e_ref)
        denom = np.abs(want) + 1e-4 * np.abs(want).max()
        el_hip, el_ref = (np.abs(N(got) - want) / denom).max(), (np.abs(ref32 - want) / denom).max()
        assert el_hip <= max(3 * el_ref, 1e-4), (name, el_hip, el_ref)


def test_projection_covars_path_and_radius_clip(ops):
    fx = garden(1500, scale_mult=3.0)
    means, quats, scales = T(fx["means"]), T(fx["quats"]), T(fx["scales"])
    viewmats, Ks = T(fx["viewmats"]), T(fx["Ks"])
    W, H = fx["width"], fx["height"]
    covars, _ = ops.quat_scale_to_covar_preci(quats, scales, compute_preci=False, triu=True)
    covars = covars.detach().requires_grad_(True)
    out_q = ops.fully_fused_projection(means, None, quats, scales, viewmats, Ks, W, H, radius_clip=3.0, near_plane=0.2,
                                       far_plane=6.0)
    out_c = ops.fully_fused_projection(means, covars, None, None, viewmats, Ks, W, H, radius_clip=3.0, near_plane=0.2,
                                       far_plane=6.0)
    assert (N(out_q[0]) == N(out_c[0])).mean() > 0.999
    valid = (N(out_q[0]) > 0) & (N(out_c[0]) > 0)
    for a, b in zip(out_q[1:4], out_c[1:4]):
        assert_close(N(a)[valid], N(b)[valid], 1e-4, 1e-4, "quat/scale vs covars")
    # oracle agreement incl. clip planes and radius_clip
    o = O.projection_fwd(fx["means"], None, fx["quats"], fx["scales"], fx["viewmats"], fx["Ks"], W, H,
                         near_plane=0.2, far_plane=6.0, radius_clip=3.0)
    assert (N(out_q[0]) == o[0]).mean() > 0.999
    assert ((N(out_q[0]) > 0) & (N(out_q[0]) <= 3)).sum() == 0
    # v_covars path
    v = torch.randn_like(out_c[3])
    (g_c,) = torch.autograd.grad((out_c[3] * v * (out_c[0] > 0)[..., None]).sum(), covars)
    vm2 = np.zeros(N(out_c[1]).shape, np.float32)
    vd = np.zeros(N(out_c[2]).shape, np.float32)
    vc = N(v) * (N(out_c[0]) > 0)[..., None]
    o_b = O.projection_bwd(fx["means"], N(covars), None, None, fx["viewmats"], fx["Ks"], W, H, 0.3, "pinhole",
                           N(out_c[0]), N(out_c[3]), None, vm2, vd, vc, None)
    assert rel_l2(N(g_c), o_b[1]) < 2e-3


def test_quat_scale_to_covar_preci(ops):
    """reference tests/test_basic.py:52-88"""
    fx = garden(500)
    quats, scales = T(fx["quats"], True), T(fx["scales"] * 50 + 0.05, True)
    for triu in (False, True):
        covars, precis = ops.quat_scale_to_covar_preci(quats, scales, triu=triu)
        q, s = quats.detach().cpu().double(), scales.detach().cpu().double()
        qn = q / q.norm(dim=-1, keepdim=True)
        w, x, y, z = qn.unbind(-1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                         1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                         1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
        M = R * s[:, None, :]
        cov = M @ M.transpose(1, 2)
        P = R / s[:, None, :]
        pre = P @ P.transpose(1, 2)
        if triu:
            idx = ([0, 0, 0, 1, 1, 2], [0, 1, 2, 1, 2, 2])
            cov, pre = cov[:, idx[0], idx[1]], pre[:, idx[0], idx[1]]
        assert_close(N(covars), cov.numpy(), 1e-4, 1e-6, "covars")
        assert_close(N(precis), pre.numpy(), 1e-3, 1e-3, "precis")
        vc, vp = torch.randn_like(covars), torch.randn_like(precis) * 1e-3
        g_q, g_s = torch.autograd.grad((covars * vc).sum() + (precis * vp).sum(), (quats, scales))
        qd, sd = q.clone().requires_grad_(True), s.clone().requires_grad_(True)
        qn = qd / qd.norm(dim=-1, keepdim=True)
        w, x, y, z = qn.unbind(-1)
        R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z),
                         1 - 2 * (x * x + z * z), 2 * (y * z - w * x), 2 * (x * z - w * y), 2 * (y * z + w * x),
                         1 - 2 * (x * x + y * y)], -1).reshape(-1, 3, 3)
        M = R * sd[:, None, :]
        cov = M @ M.transpose(1, 2)
        P = R / sd[:, None, :]
        pre = P @ P.transpose(1, 2)
        if triu:
            cov, pre = cov[:, idx[0], idx[1]], pre[:, idx[0], idx[1]]
        r_q, r_s = torch.autograd.grad((cov * vc.cpu().double()).sum() + (pre * vp.cpu().double()).sum(), (qd, sd))
        assert rel_l2(N(g_q), r_q.numpy()) < 1e-3
        assert rel_l2(N(g_s), r_s.numpy()) < 1e-3


# ---------------------------------------------------------------------------
# spherical harmonics  (reference tests/test_basic.py:579-607, tol 1e-4)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("sh_degree", [0, 1, 2, 3, 4])
def test_sh_vs_golden(ops, sh_degree):
    gd = golden("sh.npz")
    coeffs, dirs = T(gd["coeffs"], True), T(gd["dirs"], True)
    colors = ops.spherical_harmonics(sh_degree, dirs, coeffs)
    assert_close(N(colors), gd[f"deg{sh_degree}_colors"], 1e-4, 1e-4, "colors")
    g_c, g_d = torch.autograd.grad((colors * T(gd["v_colors"])).sum(), (coeffs, dirs), allow_unused=True)
    assert_close(N(g_c), gd[f"deg{sh_degree}_v_coeffs"], 1e-4, 1e-4, "v_coeffs")
    if sh_degree > 0:
        assert_close(N(g_d), gd[f"deg{sh_degree}_v_dirs"], 1e-4, 1e-4, "v_dirs")


@pytest.mark.parametrize("K", [16, 25, 9])
def test_sh_masks_shared_and_partial_bands(ops, K):
    rs = np.random.RandomState(1)
    C, Ng, deg = 3, 777, 2
    dirs = rs.randn(C, Ng, 3).astype(np.float32)
    coeffs = rs.randn(Ng, K, 3).astype(np.float32)
    masks = rs.rand(C, Ng) > 0.4
    v = rs.randn(C, Ng, 3).astype(np.float32)
    d_t, c_t = T(dirs, True), T(coeffs, True)
    col = ops.spherical_harmonics_shared(deg, d_t, c_t, masks=T(masks))
    g_c, g_d = torch.autograd.grad((col * T(v) * T(masks)[..., None]).sum(), (c_t, d_t))
    # oracle on the materialised [C,N,K,3] form (what the reference does)
    cexp = np.ascontiguousarray(np.broadcast_to(coeffs[None], (C, Ng, K, 3)))
    o_col = O.sh_fwd(deg, dirs, cexp, masks)
    o_vc, o_vd = O.sh_bwd(deg, dirs, cexp, v * masks[..., None], masks)
    assert_close(N(col)[masks], o_col[masks], 1e-4, 1e-5, "colors (masked)")
    assert_close(N(g_c), o_vc.sum(0), 1e-4, 1e-5, "v_coeffs summed over cameras")
    assert_close(N(g_d), o_vd, 1e-4, 1e-5, "v_dirs")
    # inactive bands of the gradient are exactly zero
    assert (N(g_c)[:, (deg + 1) ** 2:] == 0).all()
    # generic (non-shared) entry point on the expanded tensor gives the same numbers
    c2 = T(cexp, True)
    col2 = ops.spherical_harmonics(deg, T(dirs), c2, masks=T(masks))
    assert_close(N(col2)[masks], N(col)[masks], 0, 0, "shared vs expanded")
    (g_c2,) = torch.autograd.grad((col2 * T(v) * T(masks)[..., None]).sum(), (c2,))
    assert_close(N(g_c2), o_vc, 1e-4, 1e-5, "v_coeffs per camera")


# ---------------------------------------------------------------------------
# tile intersection + sort + offsets: BIT EXACT  (reference tests/test_basic.py:442-472)
# ---------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["a", "b"])
def test_isect_bit_exact_vs_reference_golden(ops, case):
    gd = golden("isect.npz")
    means2d, radii, depths = T(gd[f"{case}_means2d"]), T(gd[f"{case}_radii"]), T(gd[f"{case}_depths"])
    ts, tw, th = int(gd[f"{case}_tile_size"]), int(gd[f"{case}_tile_width"]), int(gd[f"{case}_tile_height"])
    C = means2d.shape[0]
    tpg, ids, flat = ops.isect_tiles(means2d, radii, depths, ts, tw, th)
    offs = ops.isect_offset_encode(ids, C, tw, th)
    assert tpg.dtype == torch.int32 and ids.dtype == torch.int64 and flat.dtype == torch.int32 and offs.dtype == torch.int32
    assert np.array_equal(N(tpg), gd[f"{case}_tiles_per_gauss"])
    assert np.array_equal(N(ids), gd[f"{case}_isect_ids"])
    assert np.array_equal(N(flat), gd[f"{case}_flatten_ids"])
    assert np.array_equal(N(offs), gd[f"{case}_isect_offsets"])
    if case == "a":
        _, ids_u, flat_u = ops.isect_tiles(means2d, radii, depths, ts, tw, th, sort=False)
        assert np.array_equal(N(ids_u), gd["a_isect_ids_unsorted"])
        assert np.array_equal(N(flat_u), gd["a_flatten_ids_unsorted"])


@pytest.mark.parametrize("C,tw,th,ts,n", [(5, 256, 256, 16, 20000), (1, 1024, 64, 4, 30000), (9, 3, 2, 16, 5000), (64, 8, 8, 16, 300),
                                          (2, 2048, 1024, 16, 4000)])
def test_isect_key_bit_budget(ops, C, tw, th, ts, n):
    """The key layout at the ends of its range (reference isect_tiles.cu:155-159: tile_n_bits + cam_n_bits <= 32): up to
    21 tile bits and 7 camera bits, i.e. one to three pair-sort passes and every split of the leading digit, bit-exact
    against the oracle, with depth ties (stability) and splats hanging over the image border."""
    rng = np.random.default_rng(C * 1000 + tw)
    W, H = tw * ts, th * ts
    means2d = (rng.random((C, n, 2), dtype=np.float32) * np.array([W + 40, H + 40], np.float32) - 20).astype(np.float32)
    radii = rng.integers(0, 3 * ts, (C, n)).astype(np.int32)
    radii[rng.random((C, n)) < 0.3] = 0
    depths = np.round(rng.random((C, n), dtype=np.float32) * 8 + 0.5, 1).astype(np.float32)  # many exact ties
    tpg, ids, flat = ops.isect_tiles(T(means2d), T(radii), T(depths), ts, tw, th)
    offs = ops.isect_offset_encode(ids, C, tw, th)
    o_tpg, o_ids, o_flat = O.isect_tiles(means2d, radii, depths, ts, tw, th)
    o_offs = O.isect_offset_encode(o_ids, C, tw, th)
    assert np.array_equal(N(tpg), o_tpg)
    assert np.array_equal(N(ids), o_ids)
    assert np.array_equal(N(flat), o_flat)
    assert np.array_equal(N(offs), o_offs)


def test_isect_packed_and_empty(ops):
    gd = golden("isect.npz")
    m, r, d = gd["b_means2d"], gd["b_radii"], gd["b_depths"]
    C, Ng = r.shape
    ts, tw, th = 16, int(gd["b_tile_width"]), int(gd["b_tile_height"])
    cam, gau = np.nonzero(r > 0)
    tpg, ids, flat = ops.isect_tiles(T(m[cam, gau]), T(r[cam, gau]), T(d[cam, gau]), ts, tw, th, packed=True,
                                     n_cameras=C, camera_ids=T(cam.astype(np.int64)), gaussian_ids=T(gau.astype(np.int64)))
    o_tpg, o_ids, o_flat = O.isect_tiles(m[cam, gau], r[cam, gau], d[cam, gau], ts, tw, th, n_cameras=C,
                                         camera_ids=cam.astype(np.int64))
    assert np.array_equal(N(tpg), o_tpg) and np.array_equal(N(ids), o_ids) and np.array_equal(N(flat), o_flat)
    # same ids as the unpacked run (flatten ids differ by construction)
    assert np.array_equal(N(ids), gd["b_isect_ids"])
    # nothing visible -> empty lists, zero offsets
    z = torch.zeros((2, 10), dtype=torch.int32, device=T(m).device)
    tpg, ids, flat = ops.isect_tiles(torch.zeros((2, 10, 2), device=z.device), z, torch.ones((2, 10), device=z.device), 16, 4, 4)
    assert ids.numel() == 0 and flat.numel() == 0 and int(tpg.sum()) == 0
    offs = ops.isect_offset_encode(ids, 2, 4, 4)
    assert int(offs.abs().sum()) == 0


@pytest.mark.parametrize("n,end_bit", [(1, 40), (63, 46), (4096, 46), (4097, 33), (200_003, 46), (1_000_000, 64), (300_000, 8)])
def test_radix_sort_stable_bit_exact(ops, n, end_bit):
    from gscodec_studio_amd import _backend as B

    rs = np.random.RandomState(n % 1000)
    # few distinct keys -> many ties -> stability is actually exercised
    keys = rs.randint(0, 1 << 20, size=n).astype(np.int64) << 12
    keys |= rs.randint(0, 4, size=n).astype(np.int64) << 44
    if end_bit == 64:
        keys |= rs.randint(0, 2, size=n).astype(np.int64) << 63  # negative int64 keys
    vals = np.arange(n, dtype=np.int32)
    k_t, v_t = T(keys), T(vals)
    ko, vo = torch.empty_like(k_t), torch.empty_like(v_t)
    tb = B.query("gs_sort_temp_bytes", n)
    temp = torch.empty(tb, dtype=torch.uint8, device=k_t.device)
    B.call("gs_sort_pairs_u64_i32", n, B.ptr(k_t), B.ptr(v_t), B.ptr(ko), B.ptr(vo), 0, end_bit, B.ptr(temp), tb,
           torch.cuda.current_stream().cuda_stream)
    ek, ev = O.sort_pairs(keys, vals, end_bit)
    assert np.array_equal(N(k_t), keys), "inputs must not be modified"
    assert np.array_equal(N(ko), ek)
    assert np.array_equal(N(vo), ev)
    # independent check: numpy stable sort on the masked (signed when 64 bits) key
    mk = keys if end_bit == 64 else (keys & ((1 << end_bit) - 1))
    order = np.argsort(mk, kind="stable")
    assert np.array_equal(N(vo), vals[order])


@pytest.mark.parametrize("n", [1, 1000, 4097, 300_000, 2_200_000, 3_000_001])
def test_radix_sort_drop_variant_and_gather_cumsum(ops, n):
    """gs_sort_pairs_u64_i32_drop: keys with a given upper word vanish in the first pass, the survivors come out in stable
    sorted order, their count lands on the device; gs_cumsum_gather_i32 honours that device-side count.  Sizes on both
    sides of the 1024- / 4096-key block switch (2 M keys)."""
    from gscodec_studio_amd import _backend as B

    rs = np.random.RandomState(n % 1009)
    hi = rs.randint(0, 1 << 10, size=n).astype(np.int64) << 21  # few distinct values: ties
    drop = rs.rand(n) < 0.7
    hi[drop] = 0x7FFFFFFF
    keys = (hi << 32) | np.arange(n, dtype=np.int64)
    vals = np.arange(n, dtype=np.int32)
    k_t, v_t = T(keys), T(vals)
    ko, vo = torch.full_like(k_t, -1), torch.full_like(v_t, -1)
    n_kept = torch.zeros(1, dtype=torch.int32, device=k_t.device)
    tb = B.query("gs_sort_temp_bytes", n)
    temp = torch.empty(tb, dtype=torch.uint8, device=k_t.device)
    st = torch.cuda.current_stream().cuda_stream
    # side sums: the last pass also leaves sum(src[value]) per group of 2^shift output positions behind
    src = rs.randint(0, 9, size=n).astype(np.int32)
    shift = int(B.query("gs_isect_emit_group_shift"))
    side = torch.full(((n + (1 << shift) - 1) >> shift,), 12345, dtype=torch.int32, device=k_t.device)  # (zero-filled by the call)
    src_t = T(src)
    B.call("gs_sort_pairs_u64_i32_drop", n, B.ptr(k_t), B.ptr(v_t), B.ptr(ko), B.ptr(vo), 32, 64, 0x7FFFFFFF, B.ptr(n_kept),
           B.ptr(temp), tb, 0, B.ptr(src_t), B.ptr(side), shift, st)
    kept = np.nonzero(~drop)[0]
    assert int(n_kept.item()) == len(kept)
    order = kept[np.argsort(keys[kept] >> 32, kind="stable")]
    assert np.array_equal(N(vo)[: len(kept)], vals[order]) and np.array_equal(N(ko)[: len(kept)], keys[order])
    assert np.all(N(vo)[len(kept):] == -1)  # the rest of the outputs is untouched
    padded = np.zeros(side.numel() << shift, np.int64)
    padded[: len(kept)] = src[order]
    assert np.array_equal(N(side).astype(np.int64), padded.reshape(-1, 1 << shift).sum(1))
    # the same sort without side sums (both NULL)
    ko2, vo2 = torch.full_like(k_t, -1), torch.full_like(v_t, -1)
    B.call("gs_sort_pairs_u64_i32_drop", n, B.ptr(k_t), B.ptr(v_t), B.ptr(ko2), B.ptr(vo2), 32, 64, 0x7FFFFFFF, B.ptr(n_kept),
           B.ptr(temp), tb, 0, None, None, 0, st)
    assert torch.equal(ko2, ko) and torch.equal(vo2, vo)
    # prefix sum of src[perm[i]] over the kept positions only
    out = torch.empty(n, dtype=torch.int64, device=k_t.device)
    sb = B.query("gs_cumsum_scratch_bytes", n)
    scratch = torch.empty(sb, dtype=torch.uint8, device=k_t.device)
    B.call("gs_cumsum_gather_i32", n, B.ptr(T(src)), B.ptr(vo), B.ptr(n_kept), B.ptr(out), B.ptr(scratch), sb, st)
    # (defined over the kept prefix; behind it the output is only written up to the end of the last scan block in use)
    assert np.array_equal(N(out)[: len(kept)], np.cumsum(src[order].astype(np.int64)))


def test_cumsum_matches_numpy(ops):
    from gscodec_studio_amd import _backend as B

    for n in (1, 255, 2048, 2049, 1_000_003, 4096 * 2048 + 5):  # the last one: more than 4096 blocks (separate spine launch)
        x = np.random.RandomState(n % 97).randint(0, 50, size=n).astype(np.int32)
        x_t = T(x)
        out = torch.empty(n, dtype=torch.int64, device=x_t.device)
        sb = B.query("gs_cumsum_scratch_bytes", n)
        scratch = torch.empty(sb, dtype=torch.uint8, device=x_t.device)
        B.call("gs_cumsum_i32", n, B.ptr(x_t), B.ptr(out), B.ptr(scratch), sb, torch.cuda.current_stream().cuda_stream)
        assert np.array_equal(N(out), np.cumsum(x.astype(np.int64)))


# ---------------------------------------------------------------------------
# compositing  (reference tests/test_basic.py:475-576)
# ---------------------------------------------------------------------------
def _raster_case(n=3000, scale_mult=6.0, cams=2, channels=3, seed=0, opac_boost=False):
    fx = garden(n, scale_mult=scale_mult)
    W, H = fx["width"], fx["height"]
    radii, means2d, depths, conics, _ = O.projection_fwd(fx["means"], None, fx["quats"], fx["scales"],
                                                         fx["viewmats"][:cams], fx["Ks"][:cams], W, H)
    rs = np.random.RandomState(seed)
    C = cams
    opac = np.broadcast_to(fx["opacities"][None], (C, n)).copy()
    if opac_boost:
        opac = np.clip(opac * 3.0, 0, 1).astype(np.float32)  # saturate pixels -> early termination + alpha clamp
    colors = rs.rand(C, n, channels).astype(np.float32)
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids, flat = O.isect_tiles(means2d, radii, depths, 16, tw, th)
    offs = O.isect_offset_encode(ids, C, tw, th)
    return dict(means2d=means2d, conics=conics, colors=colors, opacities=opac, W=W, H=H, offs=offs, flat=flat, C=C)


_R4_CASES = {(3, True), (1, True), (4, True), (7, True), (32, True), (40, True)}


# channels x absgrad: 1..4 the hot kernels; 5..16 the wide instances (8 / 9 / 12 / 16) in one launch; 17..32 two launches over
# halves without absgrad, the generic one-pass backward with it; 40: generic (round 5: 5, 8, 9, 12, 16, 17, 24 added, 9 being the
# reference's spacetime render, examples/simple_trainer_STG.py:531-551, and the no-absgrad halves of 32)
@pytest.mark.parametrize("channels,absgrad", [(3, True), (1, True), (4, True), (7, True), (32, True), (40, True), (5, False), (8, True),
                                              (9, False), (9, True), (12, False), (16, True), (16, False), (17, False), (24, False),
                                              (32, False)])
def test_rasterize_fwd_bwd_vs_oracle(ops, channels, absgrad):
    c = _raster_case(n=2500 if channels > 8 else 4000, channels=channels, opac_boost=(channels in (3, 9, 32)))
    rs = np.random.RandomState(5)
    bg = rs.rand(c["C"], channels).astype(np.float32)
    o_rc, o_ra, o_li, bl = O.rasterize_fwd(c["means2d"], c["conics"], c["colors"], c["opacities"], c["W"], c["H"], 16,
                                           c["offs"], c["flat"], backgrounds=bg, return_borderline=True)
    m2, cn, col, op = T(c["means2d"], True), T(c["conics"], True), T(c["colors"], True), T(c["opacities"], True)
    bg_t = T(bg, True)
    rc, ra = ops.rasterize_to_pixels(m2, cn, col, op, c["W"], c["H"], 16, T(c["offs"]), T(c["flat"]), backgrounds=bg_t,
                                     absgrad=absgrad)
    ok = bl == 0  # pixels whose threshold decisions are not within a few ulp of flipping
    r4 = (channels, absgrad) in _R4_CASES
    if not r4:  # (round-5 cases: the float64 oracle is the gradient reference below; a pixel IT flags is borderline too)
        with O.precision(64):
            _, d_ra, d_li, bl64 = O.rasterize_fwd(c["means2d"], c["conics"], c["colors"], c["opacities"], c["W"], c["H"], 16,
                                                  c["offs"], c["flat"], backgrounds=bg, return_borderline=True)
        ok = ok & (bl64 == 0)
    assert ok.mean() > 0.995
    assert_close(N(rc)[ok], o_rc[ok], 1e-4, 2e-5, "render_colors", max_bad_frac=2e-5)
    assert_close(N(ra)[ok], o_ra[ok], 1e-4, 2e-5, "render_alphas", max_bad_frac=2e-5)

    v_rc = rs.randn(*o_rc.shape).astype(np.float32) * ok[..., None]
    v_ra = rs.randn(*o_ra.shape).astype(np.float32) * ok[..., None]
    loss = (rc * T(v_rc)).sum() + (ra * T(v_ra)).sum()
    g_m2, g_cn, g_col, g_op, g_bg = torch.autograd.grad(loss, (m2, cn, col, op, bg_t))
    # oracle backward from the oracle's own forward state
    # (the cases added in round 5 take the FLOAT64 build of the oracle as their reference: one entry in 8000 of the fp32 oracle's
    # own v_opacities sat 0.4 % off at 8 channels -- its error bar, not the kernels': HIP 2.8e-5, oracle 2.3e-4 in max norm)
    with O.precision(32 if r4 else 64):
        o = O.rasterize_bwd(c["means2d"], c["conics"], c["colors"], c["opacities"], c["W"], c["H"], 16, c["offs"],
                            c["flat"], o_ra if r4 else d_ra, o_li if r4 else d_li, v_rc, v_ra, backgrounds=bg, absgrad=True)
    # Tolerances here are those of the fp32 ORACLE, not of the kernels: against a float64 ground truth the oracle's own
    # gradients are 4e-5 ... 2e-4 off (rel. L2) and the HIP kernels 2e-6 ... 3e-5 (tests/test_gpu_parity_f64.py, which
    # carries the 1e-4 assertions); this test pins decisions, plumbing, channel counts and absgrad.
    for name, got, ref in (("v_means2d", g_m2, o[0]), ("v_conics", g_cn, o[1]), ("v_colors", g_col, o[2]),
                           ("v_opacities", g_op, o[3])) + ((("absgrad", m2.absgrad, o[4]),) if absgrad else ()):
        assert rel_l2(N(got), ref) < 2e-4, (name, rel_l2(N(got), ref))
        assert_close(N(got), ref, 1e-3, 1e-4 * np.abs(ref).max(), name, max_bad_frac=1e-4)
    o_vbg = (v_rc * (1.0 - o_ra)).sum(axis=(1, 2))
    assert_close(N(g_bg), o_vbg, 1e-3, 1e-3, "v_backgrounds")


@pytest.mark.parametrize("channels,absgrad", [(9, False), (16, True), (24, False)])
def test_wide_backward_gradient_layouts_through_the_c_abi(ops, channels, absgrad):
    """gs_rasterize_bwd with a plan + scratch (segmented wide kernels) takes the gradients three ways: `packed16 = 2` (geometry rows
    + dense colour array: what rasterize_to_pixels uses), `packed16 = 0` (the reference's four separate arrays: a C-ABI caller's
    form, reachable from no Python wrapper) and, without scratch, the generic kernels.  All three must agree."""
    import ctypes

    from gscodec_studio_amd import _backend as B
    from gscodec_studio_amd import _wrapper as W

    c = _raster_case(n=2500, channels=channels, opac_boost=True)
    C, H, Wd = c["C"], c["H"], c["W"]
    dev = torch.device("cuda")
    m2, cn, col, op = T(c["means2d"]), T(c["conics"]), T(c["colors"]), T(c["opacities"])
    offs, flat = T(c["offs"]), T(c["flat"])
    th, tw = c["offs"].shape[1:]
    n_elems, n_isects = op.numel(), flat.shape[0]
    rs = np.random.RandomState(channels)
    v_rc, v_ra = T(rs.randn(C, H, Wd, channels).astype(np.float32)), T(rs.randn(C, H, Wd, 1).astype(np.float32))
    st = torch.cuda.current_stream().cuda_stream
    plan, sb = W._raster_plan(C * th * tw, n_isects, channels)
    scratch = torch.empty(sb, dtype=torch.uint8, device=dev)
    rc = torch.empty((C, H, Wd, channels), device=dev)
    ra = torch.empty((C, H, Wd, 1), device=dev)
    li = torch.empty((C, H, Wd), dtype=torch.int32, device=dev)
    B.call("gs_rasterize_fwd", C, n_elems, n_isects, channels, B.ptr(m2), B.ptr(cn), B.ptr(col), B.ptr(op), None, None, None, Wd, H, 16, tw, th,
           B.ptr(offs), B.ptr(flat), B.ptr(rc), B.ptr(ra), B.ptr(li), ctypes.addressof(plan), B.ptr(scratch), None, 0, st)

    def bwd(packed16, use_plan):
        z = lambda *shape: torch.zeros(shape, device=dev)  # noqa: E731
        if packed16 == 2:
            P, vcol = z(C, c["means2d"].shape[1], 16), z(*col.shape)
            ptrs = (B.ptr(P) if absgrad else None, B.ptr(P), None, B.ptr(vcol), None)
        else:
            vm, vc_, vcol, vo, va = z(*m2.shape), z(*cn.shape), z(*col.shape), z(*op.shape), (z(*m2.shape) if absgrad else None)
            ptrs = (B.ptr(va), B.ptr(vm), B.ptr(vc_), B.ptr(vcol), B.ptr(vo))
        B.call("gs_rasterize_bwd", C, n_elems, n_isects, channels, B.ptr(m2), B.ptr(cn), B.ptr(col), B.ptr(op), None, None, None, Wd, H, 16, tw,
               th, B.ptr(offs), B.ptr(flat), B.ptr(rc), B.ptr(ra), B.ptr(li), B.ptr(v_rc), B.ptr(v_ra), channels, 1, *ptrs, packed16, None,
               ctypes.addressof(plan) if use_plan else None, B.ptr(scratch) if use_plan else None, st)
        if packed16 == 2:
            return P[..., 0:2], P[..., 2:5], vcol, P[..., 5], (P[..., 10:12] if absgrad else None)
        return vm, vc_, vcol, vo, va

    a, b, g = bwd(2, True), bwd(0, True), bwd(0, False)
    for name, x, y, z_ in zip(("v_means2d", "v_conics", "v_colors", "v_opacities", "absgrad"), a, b, g):
        if x is None:
            continue
        assert rel_l2(N(y), N(x)) < 2e-5, (name, "separate arrays vs rows", rel_l2(N(y), N(x)))
        assert rel_l2(N(z_), N(x)) < 2e-4, (name, "generic vs segmented", rel_l2(N(z_), N(x)))


def test_rasterize_masks_tilesize_and_last_ids(ops):
    c = _raster_case(n=2000, cams=1, channels=3)
    th, tw = c["offs"].shape[1:]
    rs = np.random.RandomState(2)
    masks = rs.rand(1, th, tw) > 0.3
    o_rc, o_ra, o_li = O.rasterize_fwd(c["means2d"], c["conics"], c["colors"], c["opacities"], c["W"], c["H"], 16,
                                       c["offs"], c["flat"], masks=masks)
    rc, ra = ops.rasterize_to_pixels(T(c["means2d"]), T(c["conics"]), T(c["colors"]), T(c["opacities"]), c["W"], c["H"],
                                     16, T(c["offs"]), T(c["flat"]), masks=T(masks))
    pm = np.repeat(np.repeat(masks, 16, 1), 16, 2)[:, :c["H"], :c["W"]]
    assert_close(N(rc)[pm], o_rc[pm], 1e-4, 2e-5, "masked render (kept tiles)", max_bad_frac=1e-4)
    assert (N(rc)[~pm] == 0).all()
    # ... and their backward: masked tiles contribute nothing (their workgroups leave the forward early, but still clear their
    # share of the gradient rows)
    m2, cn, col, op = T(c["means2d"], True), T(c["conics"], True), T(c["colors"], True), T(c["opacities"], True)
    rc2, ra2 = ops.rasterize_to_pixels(m2, cn, col, op, c["W"], c["H"], 16, T(c["offs"]), T(c["flat"]), masks=T(masks))
    o_rc, o_ra, o_li, bl = O.rasterize_fwd(c["means2d"], c["conics"], c["colors"], c["opacities"], c["W"], c["H"], 16,
                                           c["offs"], c["flat"], masks=masks, return_borderline=True)
    ok = (bl == 0)
    v_rc = rs.randn(*o_rc.shape).astype(np.float32) * ok[..., None]
    v_ra = rs.randn(*o_ra.shape).astype(np.float32) * ok[..., None]
    g = torch.autograd.grad((rc2 * T(v_rc)).sum() + (ra2 * T(v_ra)).sum(), (m2, cn, col, op))
    o = O.rasterize_bwd(c["means2d"], c["conics"], c["colors"], c["opacities"], c["W"], c["H"], 16, c["offs"], c["flat"],
                        o_ra, o_li, v_rc, v_ra, masks=masks)
    for name, got, ref in zip(("v_means2d", "v_conics", "v_colors", "v_opacities"), g, o[:4]):
        assert rel_l2(N(got), ref) < 2e-4, (name, rel_l2(N(got), ref))
    # small tiles
    fx = garden(1500, scale_mult=6.0)
    W, H = 160, 100
    Ks = fx["Ks"][:1].copy()
    Ks[:, 0] *= W / fx["width"]
    Ks[:, 1] *= H / fx["height"]
    radii, means2d, depths, conics, _ = O.projection_fwd(fx["means"], None, fx["quats"], fx["scales"], fx["viewmats"][:1], Ks, W, H)
    for ts in (4, 8, 13):
        tw, th = math.ceil(W / ts), math.ceil(H / ts)
        tpg, ids, flat = O.isect_tiles(means2d, radii, depths, ts, tw, th)
        offs = O.isect_offset_encode(ids, 1, tw, th)
        cols = rs.rand(1, 1500, 3).astype(np.float32)
        op = fx["opacities"][None].copy()
        o_rc, o_ra, o_li, bl = O.rasterize_fwd(means2d, conics, cols, op, W, H, ts, offs, flat, return_borderline=True)
        rc, ra = ops.rasterize_to_pixels(T(means2d), T(conics), T(cols), T(op), W, H, ts, T(offs), T(flat))
        assert_close(N(rc)[bl == 0], o_rc[bl == 0], 1e-4, 2e-5, f"tile {ts}", max_bad_frac=1e-4)


# --------------------------------------------------------------------------- exchange row packing
@pytest.mark.parametrize("n_rows", [0, 1, 255, 256, 257, 70001])
def test_rows_pack_unpack(n_rows):
    """gs_rows_pack / gs_rows_unpack against torch.cat / column slices: dense parts, an int32 part travelling as its bit
    pattern, a missing part (zeros), and parts that are column views of a wider buffer (read in place)."""
    from gscodec_studio_amd._wrapper import rows_pack, rows_unpack

    g = torch.Generator(device="cuda").manual_seed(n_rows)
    radii = torch.randint(-5, 1000, (n_rows,), device="cuda", dtype=torch.int32, generator=g)
    m2 = torch.randn(n_rows, 2, device="cuda", generator=g)
    wide = torch.randn(n_rows, 16, device="cuda", generator=g)  # packed gradient rows: conic at 4..6, opacity at 10
    conic, opac = wide[:, 4:7], wide[:, 10]
    col = torch.randn(n_rows, 5, device="cuda", generator=g)
    parts = [(radii, 1), (m2, 2), (None, 1), (conic, 3), (opac.reshape(1, -1), 1), (col, 5)]
    wire = rows_pack(parts, n_rows, m2)
    want = torch.cat([radii.view(torch.float32)[:, None], m2, torch.zeros(n_rows, 1, device="cuda"), conic, opac[:, None], col], dim=1)
    assert wire.shape == (n_rows, 13)
    assert torch.equal(wire.view(torch.int32), want.view(torch.int32))
    outs = [torch.empty(n_rows, dtype=torch.int32, device="cuda"), torch.empty(n_rows, 2, device="cuda"), None,
            torch.empty(n_rows, 3, device="cuda"), torch.empty(1, n_rows, device="cuda"), torch.empty(n_rows, 5, device="cuda")]
    rows_unpack(wire, list(zip(outs, [1, 2, 1, 3, 1, 5])))
    assert torch.equal(outs[0], radii) and torch.equal(outs[1], m2) and torch.equal(outs[3], conic)
    assert torch.equal(outs[4][0], opac) and torch.equal(outs[5], col)


@pytest.mark.parametrize("width", [64, 70, 523])
def test_rows_pack_unpack_wide_rows(width):
    """Wire rows wider than 64 elements (the exchange of D >= 55 colour channels; the reference's distributed path takes any
    D): fewer rows per workgroup, same result."""
    from gscodec_studio_amd._wrapper import rows_pack, rows_unpack

    n_rows = 3001
    g = torch.Generator(device="cuda").manual_seed(width)
    hdr = torch.randint(0, 1 << 20, (n_rows, 2), device="cuda", dtype=torch.int32, generator=g)
    m2 = torch.randn(n_rows, 2, device="cuda", generator=g)
    col = torch.randn(n_rows, width - 4, device="cuda", generator=g)
    wire = rows_pack([(hdr, 2), (m2, 2), (col, width - 4)], n_rows, m2)
    want = torch.cat([hdr.view(torch.float32), m2, col], dim=1)
    assert wire.shape == (n_rows, width) and torch.equal(wire.view(torch.int32), want.view(torch.int32))
    o_h, o_m, o_c = torch.empty_like(hdr), torch.empty_like(m2), torch.empty_like(col)
    rows_unpack(wire, [(o_h, 2), (o_m, 2), (o_c, width - 4)])
    assert torch.equal(o_h, hdr) and torch.equal(o_m, m2) and torch.equal(o_c, col)
    idx = torch.randperm(n_rows, device="cuda", generator=g)[:1000].to(torch.int32)
    wire_i = rows_pack([(idx, 1), (col, width - 4, True)], 1000, col, idx)
    assert torch.equal(wire_i[:, 1:], col[idx.long()])


def test_rows_pack_unpack_indexed():
    """The sparse form: wire row r <-> row index[r] of the indexed parts (gather in, scatter out), dense parts unaffected,
    and the index list may be a column of the wire itself."""
    from gscodec_studio_amd._wrapper import rows_pack, rows_unpack

    g = torch.Generator(device="cuda").manual_seed(7)
    n_src, n_rows = 5000, 1777
    idx = torch.randperm(n_src, device="cuda", generator=g)[:n_rows].to(torch.int32)
    a = torch.randn(n_src, 3, device="cuda", generator=g)
    wide = torch.randn(n_src, 16, device="cuda", generator=g)
    b = wide[:, 10]  # one column of a wider buffer
    r = torch.randint(0, 99, (n_src,), device="cuda", dtype=torch.int32, generator=g)
    wire = rows_pack([(idx, 1), (a, 3, True), (b.reshape(1, -1), 1, True), (r, 1, True), (None, 2, True)], n_rows, a, idx)
    li = idx.long()
    want = torch.cat([idx.view(torch.float32)[:, None], a[li], b[li][:, None], r[li].view(torch.float32)[:, None],
                      torch.zeros(n_rows, 2, device="cuda")], dim=1)
    assert torch.equal(wire.view(torch.int32), want.view(torch.int32))
    out_a = torch.full((n_src, 3), -1.0, device="cuda")
    out_r = torch.zeros(n_src, dtype=torch.int32, device="cuda")
    rows_unpack(wire, [(None, 1), (out_a, 3, True), (None, 1), (out_r, 1, True), (None, 2)], wire[:, 0].view(torch.int32))
    ref_a = torch.full((n_src, 3), -1.0, device="cuda")
    ref_a[li] = a[li]
    ref_r = torch.zeros(n_src, dtype=torch.int32, device="cuda")
    ref_r[li] = r[li]
    assert torch.equal(out_a, ref_a) and torch.equal(out_r, ref_r)


def test_exchange_compact_lists_and_overflow_flag():
    """gs_exchange_compact: every visible (camera, gaussian) row is listed once in the chunk of its destination rank with
    the right destination row; when ONE chunk is too small, EVERY header carries the overflow flag (a receiver only sees
    the chunks addressed to it, and all ranks have to agree on repeating the exchange)."""
    from gscodec_studio_amd._wrapper import exchange_compact

    g = torch.Generator(device="cuda").manual_seed(3)
    world, C_local, N, N_total, N_off = 3, 2, 5000, 17000, 4000
    radii = (torch.rand(world * C_local, N, device="cuda", generator=g) < 0.2).to(torch.int32) * 7
    radii[0:2] = ((torch.rand(2, N, device="cuda", generator=g) < 0.6).to(torch.int32) * 3)  # destination 0 wants far more rows
    want = [int((radii[d * C_local:(d + 1) * C_local] > 0).sum()) for d in range(world)]
    for cap in (max(want) + 10, (want[0] + max(want[1:])) // 2):
        src, hdr, counters, stats = exchange_compact(radii, C_local, world, cap, N_total, N_off)
        assert counters.tolist() == want and int(stats[0]) == max(want)
        over = int(any(w > cap for w in want))
        assert int(stats[1]) == over
        src, hdr = src.view(world, cap + 1), hdr.view(world, cap + 1, 2)
        for d in range(world):
            assert int(hdr[d, cap, 0]) == -1 and int(hdr[d, cap, 1]) == (min(want[d], cap) | (over << 30)), (d, cap)
            rows = src[d, :cap]
            rows = rows[rows >= 0]
            assert rows.numel() == min(want[d], cap) and rows.unique().numel() == rows.numel()
            cam, gau = rows // N, rows % N
            assert bool(((cam // C_local) == d).all()) and bool((radii.view(-1)[rows.long()] > 0).all())
            dst = hdr[d, :cap, 0]
            dst = dst[dst >= 0]
            assert torch.equal(dst.sort().values, ((cam % C_local) * N_total + N_off + gau).to(torch.int32).sort().values)


def test_rows16_gather_scatter_round_trip():
    """gs_rows16_gather / gs_rows16_scatter (the exchange of splat rows): gathered rows equal the indexed source rows bit for
    bit with the two tag ints in columns 12 / 13, a negative index gives a zero row carrying only its tag; the scatter puts
    every row with a non-negative index (read from column 12 of the wire itself) at its place, fills radii / depths from
    columns 10 / 9, and touches nothing else."""
    from gscodec_studio_amd._wrapper import ROW, ROW_DEPTH, ROW_RADIUS, rows16_gather, rows16_scatter

    g = torch.Generator(device="cuda").manual_seed(5)
    n_src, n_wire, n_dst = 7000, 3001, 9000
    src = torch.randn(n_src, ROW, device="cuda", generator=g)
    src.view(torch.int32)[:, ROW_RADIUS] = torch.randint(1, 50, (n_src,), device="cuda", generator=g, dtype=torch.int32)
    index = torch.randint(0, n_src, (n_wire,), device="cuda", generator=g, dtype=torch.int32)
    index[::7] = -1
    dst_rows = torch.randperm(n_dst, device="cuda", generator=g)[:n_wire].to(torch.int32)  # distinct destinations
    dst_rows[::7] = -1
    tag = torch.stack([dst_rows, torch.arange(n_wire, device="cuda", dtype=torch.int32)], dim=1).contiguous()
    wire = rows16_gather(n_wire, index, 1, src, tag)
    ref = torch.zeros(n_wire, ROW, device="cuda")
    ok = index >= 0
    ref[ok] = src[index[ok].long()]
    ref.view(torch.int32)[:, 12:14] = tag
    assert torch.equal(wire.view(torch.int32), ref.view(torch.int32))
    # strided index: a column of another row buffer
    wire2 = rows16_gather(n_wire, wire.view(torch.int32)[:, 13], ROW, wire, None)  # column 13 = arange: the identity gather
    assert torch.equal(wire2.view(torch.int32), wire.view(torch.int32))

    out = torch.full((n_dst, ROW), 7.0, device="cuda")
    radii = torch.full((n_dst,), -5, dtype=torch.int32, device="cuda")
    depths = torch.full((n_dst,), -1.0, device="cuda")
    rows16_scatter(n_wire, wire.view(torch.int32)[:, 12], ROW, wire, out, radii, depths)
    ref_out, ref_r, ref_d = torch.full_like(out, 7.0), torch.full_like(radii, -5), torch.full_like(depths, -1.0)
    d = dst_rows[ok].long()
    ref_out[d] = wire[ok]
    ref_r[d] = wire.view(torch.int32)[ok, ROW_RADIUS]
    ref_d[d] = wire[ok, ROW_DEPTH]
    assert torch.equal(out.view(torch.int32), ref_out.view(torch.int32)) and torch.equal(radii, ref_r) and torch.equal(depths, ref_d)


@pytest.mark.parametrize("shape", [(5000,), (5000, 3), (5000, 16, 3)])
def test_gather_rows_and_atomic_backward(shape):
    """gather_rows == src[ids] with the same gradient (ids repeat: a splat seen by several cameras)."""
    from gscodec_studio_amd._wrapper import gather_rows

    g = torch.Generator(device="cuda").manual_seed(len(shape))
    src = torch.randn(*shape, device="cuda", generator=g)
    ids = torch.randint(0, shape[0], (12345,), device="cuda", generator=g)
    w = torch.randn(12345, *shape[1:], device="cuda", generator=g)
    a = src.clone().requires_grad_(True)
    b = src.clone().requires_grad_(True)
    out = gather_rows(a, ids)
    ref = b[ids]
    assert torch.equal(out, ref)
    (out * w).sum().backward()
    (ref * w).sum().backward()
    assert torch.allclose(a.grad, b.grad, rtol=1e-5, atol=1e-5)


def test_isect_count_read_back_paths_agree(ops, monkeypatch):
    """The intersection count reaches the host in two ways: per-block sums stored straight into pinned memory (small
    inputs) or summed on the device and copied as 8 bytes (large ones; the direct form stalled the GPU at 49 M splats).
    Both must give the same lists."""
    from gscodec_studio_amd import _wrapper as W

    g = torch.Generator(device="cpu").manual_seed(11)
    C, n, tw, th, ts = 2, 70_000, 40, 30, 16
    means2d = (torch.rand(C, n, 2, generator=g) * torch.tensor([tw * ts, th * ts])).cuda()
    radii = torch.randint(-2, 40, (C, n), generator=g, dtype=torch.int32).cuda()
    depths = (torch.rand(C, n, generator=g) * 10 + 0.1).cuda()
    outs = []
    for limit in (1 << 30, 0):
        monkeypatch.setattr(W, "_PINNED_DIRECT_MAX", limit)
        tpg, ids, flat = ops.isect_tiles(means2d, radii, depths, ts, tw, th)
        outs.append((N(tpg), N(ids), N(flat)))
    assert outs[0][1].size > 0
    for a, b in zip(outs[0], outs[1]):
        assert np.array_equal(a, b)
