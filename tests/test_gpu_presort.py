"""The bucketed depth pre-sort (gs_presort_split / gs_isect_count_keys(bucket_splitters) / gs_presort_buckets) against the LSD
radix sort it replaces (gs_sort_pairs_u64_i32_drop, itself pinned bit-exact against the reference-generated isect fixtures) and
against numpy's stable argsort: the permutation, the kept count and the emission's group sums must be IDENTICAL, for the LDS
route, the global-memory route of oversized ranges (forced through the lds_capacity knob) and adversarial key distributions."""
import numpy as np
import pytest
import torch

from util import N, T, dev

pytestmark = pytest.mark.gpu


def _state(means2d, radii, depths, on, cap=0):
    from gscodec_studio_amd import _wrapper as W

    prev = dict(W._PRESORT)
    W._PRESORT.update(on=on, lds_capacity=cap)
    try:
        C, n = radii.shape
        st = W.isect_tiles_begin(means2d, radii, depths, 16, 120, 68, True, C, n, C * n, None)
        torch.cuda.synchronize()
        k = int(st["n_kept"])
        out = (N(st["perm"])[:k].copy(), k, N(st["gsums"]).copy(), N(st["tiles_per_gauss"]).copy())
        W.isect_tiles_abandon(st)
        return out
    finally:
        W._PRESORT.update(prev)


def _case(n, kind, seed, vis=0.3, C=1):
    rs = np.random.RandomState(seed)
    if kind == "uniform":
        d = rs.uniform(0.2, 9.0, (C, n))
    elif kind == "clustered":  # most splats on two fronto-parallel planes: thousands of nearly (or exactly) equal depths
        d = np.where(rs.rand(C, n) < 0.8, 2.5 + 1e-6 * rs.randn(C, n), rs.uniform(0.3, 30.0, (C, n)))
        d = np.where(rs.rand(C, n) < 0.3, 2.5, d)
    elif kind == "equal":
        d = np.full((C, n), 1.75)
    elif kind == "ascending":
        d = np.sort(rs.uniform(0.2, 9.0, (C, n)), axis=1)
    elif kind == "descending":
        d = -np.sort(-rs.uniform(0.2, 9.0, (C, n)), axis=1)
    elif kind == "periodic":  # depth correlated with the sampling stride
        d = 1.0 + (np.arange(C * n).reshape(C, n) % 128) * 0.01
    else:
        raise ValueError(kind)
    d = d.astype(np.float32)
    radii = np.where(rs.rand(C, n) < vis, rs.randint(1, 40, (C, n)), 0).astype(np.int32)
    m2 = np.stack([rs.uniform(-50, 1970, (C, n)), rs.uniform(-50, 1130, (C, n))], -1).astype(np.float32)
    return m2, radii, d


@pytest.mark.parametrize("n,kind,vis", [
    (1, "uniform", 1.0), (63, "uniform", 0.5), (1000, "uniform", 0.0), (5000, "clustered", 0.3), (100_000, "uniform", 0.3),
    (100_000, "equal", 0.9), (300_000, "clustered", 0.6), (1_006_065, "uniform", 0.29), (1_006_065, "clustered", 1.0),
    (1_006_065, "ascending", 0.3), (700_001, "descending", 0.3), (1_000_000, "periodic", 0.5), (2_000_000, "uniform", 0.2)])
def test_bucketed_presort_equals_radix_sort(n, kind, vis):
    m2, radii, d = _case(n, kind, seed=n % 1000 + len(kind), vis=vis)
    args = (T(m2), T(radii), T(d))
    ref_perm, ref_k, ref_g, ref_t = _state(*args, on=False)
    # ground truth: stable argsort of the depth bits over the visible elements
    flat_r, flat_d = radii.reshape(-1), d.reshape(-1)
    idx = np.nonzero(flat_r > 0)[0]
    want = idx[np.argsort(flat_d[idx].view(np.uint32), kind="stable")]
    assert ref_k == len(idx) and np.array_equal(ref_perm, want)
    for cap in (0, 64):  # LDS route / every range through the global-memory route
        perm, k, g, t = _state(*args, on=True, cap=cap)
        assert k == ref_k and np.array_equal(perm, want), (kind, cap)
        assert np.array_equal(t, ref_t) and np.array_equal(g, ref_g), (kind, cap)


def test_bucketed_presort_multi_camera_and_full_pipeline():
    """C = 2 (elements of both cameras interleave in depth order) and the whole isect_tiles against the radix route."""
    from gscodec_studio_amd import _wrapper as W

    m2, radii, d = _case(200_000, "clustered", seed=5, vis=0.4, C=2)
    args = (T(m2), T(radii), T(d))
    a = _state(*args, on=True)
    b = _state(*args, on=False)
    assert a[1] == b[1] and np.array_equal(a[0], b[0]) and np.array_equal(a[2], b[2])
    outs = {}
    for on in (True, False):
        prev = dict(W._PRESORT)
        W._PRESORT.update(on=on)
        try:
            outs[on] = W.isect_tiles(*args, 16, 120, 68)
        finally:
            W._PRESORT.update(prev)
    for x, y in zip(outs[True], outs[False]):
        assert torch.equal(x, y)


def _finish(means2d, radii, depths, tw, th, packed_pairs):
    from gscodec_studio_amd import _wrapper as W

    prev = W._PACKED_PAIRS
    W._PACKED_PAIRS = packed_pairs
    try:
        C, n = radii.shape
        st = W.isect_tiles_begin(means2d, radii, depths, 16, tw, th, True, C, n, C * n, None)
        return W.isect_tiles_finish(st, offsets_for=C)
    finally:
        W._PACKED_PAIRS = prev


@pytest.mark.parametrize("n,C,vis,tw,th,kind", [
    (200_000, 1, 0.4, 120, 68, "clustered"),   # 17 position bits + 13 key bits: packed
    (1_006_065, 1, 0.29, 120, 68, "uniform"),  # BASELINE config 2's shape: 19 + 13 = 32 bits exactly
    (50_000, 2, 0.9, 40, 30, "equal"),         # two cameras (1 camera bit), every depth equal: ties resolved by position
    (3000, 3, 1.0, 8, 8, "uniform"),
    (1, 1, 1.0, 120, 68, "uniform"),           # one visible element: 0 position bits
    (700_000, 1, 1.0, 120, 68, "uniform"),     # 20 position bits + 13: does not fit, the call falls back to (key, id) pairs
])
def test_packed_pairs_equal_key_value_pairs(n, C, vis, tw, th, kind):
    """gs_isect_finish_presorted with n_kept_host (pairs as ONE 32-bit word, key << pos_bits | depth rank) against the same call
    without it ((key, flatten id) pairs): identical tiles_per_gauss, isect_ids, flatten_ids and offsets."""
    m2, radii, d = _case(n, kind, seed=n % 97, vis=vis, C=C)
    m2[..., 0] *= tw * 16 / 1920.0
    m2[..., 1] *= th * 16 / 1080.0
    args = (T(m2), T(radii), T(d))
    a = _finish(*args, tw, th, True)
    b = _finish(*args, tw, th, False)
    assert a[1].numel() > 0
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    ids = N(a[1])
    assert np.all(ids[1:] >= ids[:-1])


def test_packed_kernels_directly_with_spare_position_bits():
    """gs_isect_emit_packed + gs_sort_isect_packed with MORE position bits than needed (and the minimum): same outputs as the
    (key, id) route of gs_isect_emit_presorted + gs_sort_isect_pairs."""
    from gscodec_studio_amd import _backend as B
    from gscodec_studio_amd import _wrapper as W

    m2, radii, d = _case(60_000, "clustered", seed=3, vis=0.5, C=2)
    tw, th = 30, 20
    m2[..., 0] *= tw * 16 / 1920.0
    m2[..., 1] *= th * 16 / 1080.0
    m2t, rt, dt = T(m2), T(radii), T(d)
    C, n = radii.shape
    st = W.isect_tiles_begin(m2t, rt, dt, 16, tw, th, True, C, n, C * n, None)
    want = W.isect_tiles_finish(dict(st), offsets_for=None)  # (key, id) pairs through the separate entry points
    n_isects, n_kept = want[1].numel(), int(st["n_kept"])
    stream = torch.cuda.current_stream().cuda_stream
    key_bits = st["tile_n_bits"] + 1  # 10 tile bits + 1 bit for camera index 1
    need = max(n_kept - 1, 1).bit_length()
    for pos_bits, sk in ((need, None), (need + 3, B.ptr(st["sorted_keys"])), (32 - key_bits, None), (need, B.ptr(st["sorted_keys"]))):
        words = torch.empty(n_isects, dtype=torch.int32, device=dev())
        ids = torch.empty(n_isects, dtype=torch.int64, device=dev())
        flat = torch.empty(n_isects, dtype=torch.int32, device=dev())
        tb = B.query("gs_sort_isect_temp_bytes", n_isects)
        temp = torch.empty(tb, dtype=torch.uint8, device=dev())
        B.call("gs_isect_emit_packed", C * n, n, B.ptr(st["perm"]), B.ptr(st["n_kept"]), None, B.ptr(m2t), 2, B.ptr(rt), B.ptr(dt),
               B.ptr(st["tiles_per_gauss"]), B.ptr(st["gsums"]), B.ptr(st["gpre"]), 16, tw, th, st["tile_n_bits"], pos_bits, B.ptr(words), stream)
        B.call("gs_sort_isect_packed", n_isects, B.ptr(words), B.ptr(st["perm"]), sk, B.ptr(dt), key_bits, pos_bits, B.ptr(ids), B.ptr(flat),
               B.ptr(temp), tb, stream)
        assert torch.equal(ids, want[1]) and torch.equal(flat, want[2]), pos_bits
    with pytest.raises(RuntimeError, match="32 bits"):
        B.call("gs_sort_isect_packed", n_isects, B.ptr(words), B.ptr(st["perm"]), None, B.ptr(dt), key_bits, 33 - key_bits, B.ptr(ids), B.ptr(flat),
               B.ptr(temp), tb, stream)


def test_block_sums_summed_on_the_device_route():
    """Scenes with more count blocks than _PINNED_DIRECT_MAX add the (intersections, visible) block pairs up on the device and
    copy the two totals (the route of 49 M-splat scenes): forced here with a tiny threshold, same outputs -- packed pairs included."""
    from gscodec_studio_amd import _wrapper as W

    m2, radii, d = _case(150_000, "uniform", seed=11, vis=0.5, C=1)
    args = (T(m2), T(radii), T(d))
    a = _finish(*args, 120, 68, True)
    prev = W._PINNED_DIRECT_MAX
    W._PINNED_DIRECT_MAX = 4
    try:
        b = _finish(*args, 120, 68, True)
        c = _finish(*args, 120, 68, False)
    finally:
        W._PINNED_DIRECT_MAX = prev
    for x, y, z in zip(a, b, c):
        assert torch.equal(x, y) and torch.equal(x, z)
