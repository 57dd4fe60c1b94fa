"""CPU: host-side logic that needs no kernel -- argument validation of rasterization()
(reference rendering.py:229-277), the simulation hook tables (reference simulation.py:30-59,
527-571), the workload generator and the camera sharding helpers."""
import numpy as np
import pytest
import torch

import gscodec_studio_amd as g


def _args(N=10, C=2):
    return dict(means=torch.randn(N, 3), quats=torch.randn(N, 4), scales=torch.rand(N, 3), opacities=torch.rand(N),
                colors=torch.rand(N, 3), viewmats=torch.eye(4).repeat(C, 1, 1), Ks=torch.eye(3).repeat(C, 1, 1),
                width=32, height=32)


@pytest.mark.parametrize("mutate,exc", [
    (lambda a: a.update(means=torch.randn(10, 2)), AssertionError),
    (lambda a: a.update(quats=torch.randn(10, 3)), AssertionError),
    (lambda a: a.update(opacities=torch.rand(10, 1)), AssertionError),
    (lambda a: a.update(viewmats=torch.eye(4)), AssertionError),
    (lambda a: a.update(Ks=torch.eye(3).repeat(3, 1, 1)), AssertionError),
    (lambda a: a.update(colors=torch.rand(9, 3)), AssertionError),
])
def test_rasterization_shape_validation(mutate, exc):
    a = _args()
    mutate(a)
    with pytest.raises(exc):
        g.rasterization(**a)


def test_rasterization_mode_validation():
    a = _args()
    with pytest.raises(AssertionError):
        g.rasterization(**a, render_mode="XYZ")
    a["colors"] = torch.rand(10, 4, 3)
    with pytest.raises(AssertionError):  # (sh_degree + 1)^2 <= K
        g.rasterization(**a, sh_degree=2)
    a = _args()
    with pytest.raises(AssertionError, match="sparse_grad"):
        g.fully_fused_projection(a["means"], None, a["quats"], a["scales"], a["viewmats"], a["Ks"], 32, 32, packed=False, sparse_grad=True)
    with pytest.raises(ValueError, match="Unsupported number of color channels"):
        g.rasterize_to_pixels(torch.zeros(1, 4, 2), torch.zeros(1, 4, 3), torch.zeros(1, 4, 600), torch.zeros(1, 4), 16, 16, 16,
                              torch.zeros(1, 1, 1, dtype=torch.int32), torch.zeros(0, dtype=torch.int32))
    with pytest.raises(AssertionError, match="Assert Failed"):
        g.rasterize_to_pixels(torch.zeros(1, 4, 2), torch.zeros(1, 4, 3), torch.zeros(1, 4, 3), torch.zeros(1, 4), 64, 16, 16,
                              torch.zeros(1, 1, 1, dtype=torch.int32), torch.zeros(0, dtype=torch.int32))


def test_simulation_tables_match_reference():
    from gscodec_studio_amd.compression_simulation import CompressionSimulation, STGCompressionSimulation

    s = CompressionSimulation(entropy_model_enable=False, entropy_steps={})
    assert s.simulation_option == {"means": False, "scales": True, "quats": True, "opacities": True, "sh0": True, "shN": True}
    assert s.bds["scales"] == [-10, 2] and s.bds["quats"] == [-1, 1] and s.bds["opacities"] == [-15, 15] and s.bds["sh0"] == [-2, 4]
    assert all(s.q_bitwidth[k] == 8 for k in ("scales", "quats", "opacities", "sh0"))
    d = STGCompressionSimulation(quantization_sim_type="round", entropy_steps={})
    assert d.bds["opacities"] == [-7, 7] and d.bds["colors"] == [-7.5, 7.5] and d.bds["features_dir"] == [-10, 10]
    assert [k for k, v in d.simulation_option.items() if v] == ["scales", "quats", "opacities", "colors", "features_dir", "features_time"]
    with pytest.raises(ValueError):  # entropy models need their step table (wiring is tested in test_entropy_cpu.py)
        STGCompressionSimulation(quantization_sim_type="round", entropy_model_enable=True)
    with pytest.raises(NotImplementedError):  # the hash-grid Gaussian model needs the reference's CUDA-only extension
        CompressionSimulation(entropy_model_enable=True, entropy_model_type="gaussian_model", entropy_steps={"scales": 1})
    # attributes that are not simulated come back as a fresh tensor (param + 0.), simulated ones need the GPU
    p = torch.nn.Parameter(torch.randn(5, 3))
    new, bits = s.simulate_compression({"means": p}, step=0)
    assert new["means"] is not p and torch.equal(new["means"], p) and bits["means"] is None


def test_workload_generator_is_deterministic_and_sized():
    from gscodec_studio_amd._helper import load_test_data, rescale_intrinsics, sh_workload

    a = load_test_data(device="cpu", scene_grid=1)
    b = load_test_data(device="cpu", scene_grid=1)
    assert a[0].shape == (111785, 3)
    for x, y in zip(a[:7], b[:7]):
        assert torch.equal(x, y)
    assert a[7:] == (648, 420)
    assert float(a[2].max()) <= 0.02 and float(a[3].max()) <= 1.0
    assert torch.allclose(a[1].norm(dim=-1), torch.ones(111785), atol=1e-5)
    K = rescale_intrinsics(a[6], 648, 420, 1920, 1080)
    assert torch.allclose(K[:, 0, 0], a[6][:, 0, 0] * 1920 / 648)
    w = sh_workload(scene_grid=1, n_cameras=5, device="cpu")
    assert w["sh"].shape == (111785, 16, 3) and w["viewmats"].shape == (5, 4, 4)
    assert not torch.equal(w["viewmats"][3], w["viewmats"][0])  # cameras beyond the fixture's 3 are distinct
    R = w["viewmats"][4, :3, :3]
    assert torch.allclose(R @ R.T, torch.eye(3), atol=1e-5)


def test_camera_sharding():
    from gscodec_studio_amd.distributed import shard_cameras

    assert shard_cameras(8, 3, 8) == [3]
    assert shard_cameras(8, 1, 4) == [1, 5]
    got = sorted(sum((shard_cameras(10, r, 4) for r in range(4)), []))
    assert got == list(range(10))


def test_big_tiles_are_split_into_sub_tiles_with_their_own_list_copies():
    """`_wrapper._split_big_tiles` (tile sizes 18..32 on the HIP backend: every 2s x 2s tile becomes 2 x 2 sub-tiles of s x s, each
    owning a copy of the tile's list range): offsets, lists and masks against a brute-force construction, empty tiles and an
    empty scene included."""
    from gscodec_studio_amd._wrapper import _split_big_tiles

    rs = np.random.RandomState(0)
    C, th, tw = 2, 3, 4
    counts = rs.randint(0, 5, size=(C, th, tw))
    counts[0, 1, 2] = 0
    n = int(counts.sum())
    offsets = (np.cumsum(counts.reshape(-1)) - counts.reshape(-1)).reshape(C, th, tw).astype(np.int32)
    flat = rs.randint(0, 1000, size=n).astype(np.int32)
    masks = rs.rand(C, th, tw) > 0.4
    vo, vf, vm = _split_big_tiles(torch.tensor(offsets), torch.tensor(flat), torch.tensor(masks))
    assert vo.shape == (C, 2 * th, 2 * tw) and vo.dtype == torch.int32 and vf.shape == (4 * n,) and vm.shape == (C, 2 * th, 2 * tw)
    vo_n, vf_n = vo.numpy().reshape(-1), vf.numpy()
    ends = np.append(vo_n[1:], 4 * n)
    k = 0
    for c in range(C):
        for vy in range(2 * th):
            for vx in range(2 * tw):
                a, b = offsets[c, vy // 2, vx // 2], offsets[c, vy // 2, vx // 2] + counts[c, vy // 2, vx // 2]
                assert np.array_equal(vf_n[vo_n[k]:ends[k]], flat[a:b]), (c, vy, vx)
                assert bool(vm[c, vy, vx]) == bool(masks[c, vy // 2, vx // 2])
                k += 1
    vo0, vf0, vm0 = _split_big_tiles(torch.zeros((1, 2, 2), dtype=torch.int32), torch.zeros(0, dtype=torch.int32), None)
    assert vo0.shape == (1, 4, 4) and int(vo0.abs().sum()) == 0 and vf0.numel() == 0 and vm0 is None


def test_carve_order_has_one_source():
    """distributed._CARVE_RANK (which parameter name maps onto which piece of rasterization()'s one gradient buffer) is derived from
    _wrapper.PREFILL_ORDER, the list both autograd nodes build their GradPrefill request from (round-5 advisor: the two used to be kept
    in sync by hand)."""
    from gscodec_studio_amd import _wrapper as W
    from gscodec_studio_amd import distributed as D

    for k in W.PREFILL_ORDER:
        assert k in D._CARVE_RANK, k
    order = [k for k in W.PREFILL_ORDER if k != "sh"]  # ("sh" shares the colours' slot)
    assert [D._CARVE_RANK[k] for k in order] == sorted(D._CARVE_RANK[k] for k in order)
    # a trainer's dict in ITS order comes out in the carving order
    keys = ["means", "scales", "quats", "opacities", "sh0", "shN", "motion", "trbf_scale", "omega", "trbf_center", "something_else"]
    got = [keys[i] for i in D._canonical_order(keys)]
    assert got == ["means", "quats", "scales", "opacities", "sh0", "shN", "motion", "omega", "trbf_center", "trbf_scale", "something_else"]
    # requests are sorted into that order whatever order the items are listed in
    import torch
    t = torch.zeros(3, 2)
    req = W.prefill_request((("omega", t, True), ("means", t, True), ("scales", None, True), ("quats", t, False), ("opacities", t, True)))
    assert [k for k, _ in req] == ["means", "opacities", "omega"]


def test_dynamic_slice_descriptor_logic():
    """``dynamic.DynamicSlice`` (host logic only: no kernel runs): raw / quantize bookkeeping, the bit masks and float tables the C ABI
    takes (GS_DYN_RAW_*, quant slots scales / quats / opacities / colors), the argument runs of the two entry points, refusals."""
    import ctypes

    import torch

    from gscodec_studio_amd.dynamic import DynamicSlice

    n = 5
    mo, om, c, s = torch.zeros(n, 9), torch.zeros(n, 4), torch.zeros(n, 1), torch.zeros(n, 1)
    ds = DynamicSlice.of((mo, om, c, s, 0.25))
    assert isinstance(ds, DynamicSlice) and ds.raw_mask == 0 and ds.quant_mask == 0 and ds.min_trbf is None and ds.min_trbf_arg() == -1.0
    ds.check(n)
    with pytest.raises(AssertionError):
        ds.check(n + 1)
    ds = DynamicSlice(mo, om, c, s, 0.5, raw=True, quantize={"scales": (-10, 2, 8), "colors": (-7.5, 7.5, 8)}, min_trbf=0.05)
    assert ds.raw_mask == 7 and ds.quant_mask == 0b1001 and ds.min_trbf_arg() == 0.05
    lo, hi, rng, qn = ds._tables
    assert (lo[0], hi[0], rng[0]) == (-10.0, 2.0, 12.0) and abs(qn[0] - 1 / 255) < 1e-9 and (lo[3], hi[3], rng[3]) == (-7.5, 7.5, 15.0)
    assert (lo[1], hi[1], rng[1], qn[1]) == (0.0, 0.0, 1.0, 1.0)  # unhooked slot: neutral
    bwd = ds.c_args((mo, om, c.reshape(-1), s.reshape(-1)))
    fwd = ds.c_args((mo, om, c.reshape(-1), s.reshape(-1)), fwd_on=torch.device("cpu"), N=n)
    assert len(bwd) == 11 and len(fwd) == 13 and fwd[4] == 0.5 and fwd[5] == 0.05 and bwd[5] == 7 and bwd[6] == 0b1001
    assert ds.t_vis_mask is not None and ds.t_vis_mask.shape == (n,) and ds.t_vis_mask.dtype == torch.bool
    with pytest.raises(AssertionError):  # a quantized attribute with an activation works on the raw parameter
        DynamicSlice(mo, om, c, s, 0.5, raw=("trbf_scale",), quantize={"opacities": (-7, 7, 8)})
    with pytest.raises(AssertionError):
        DynamicSlice(mo, om, c, s, 0.5, quantize={"means": (-1, 1, 8)})
    with pytest.raises(RuntimeError):  # quantized colours that do not reach the projection (more than three channels)
        DynamicSlice(mo, om, c, s, 0.5, quantize={"colors": (-7.5, 7.5, 8)}).bind(torch.zeros(n, 4), torch.zeros(n, 3), torch.zeros(n), None,
                                                                                    mo, om, c, s)


def test_block_sum_totals_is_exact():
    """``_wrapper.block_sum_totals`` (the host's reduction of the count kernel's per-block (intersections, visible) pairs, read as int64
    words) against the plain column sums, including totals beyond 2^32."""
    import numpy as np

    from gscodec_studio_amd._wrapper import block_sum_totals

    rng = np.random.default_rng(0)
    for n, hi_val in ((1, 10), (3930, 5000), (2048, 2_000_000), (2048, 2_100_000_000)):
        a = rng.integers(0, hi_val, size=2 * n, dtype=np.int64).astype(np.int32)
        want = a.reshape(-1, 2).sum(0, dtype=np.int64)
        assert block_sum_totals(a) == (int(want[0]), int(want[1])), (n, hi_val)


def test_reorder_splats_moves_parameters_optimizer_state_and_strategy_state_together():
    """compression.reorder_splats: one permutation for every per-gaussian tensor of a trainer (parameters, Adam moments, the strategy's
    running statistics); training continues as if nothing happened -- the same updates, permuted."""
    from gscodec_studio_amd.compression import morton_order, reorder_splats

    g = torch.Generator().manual_seed(3)
    n = 257

    def make():
        params = torch.nn.ParameterDict({"means": torch.nn.Parameter(torch.randn(n, 3, generator=torch.Generator().manual_seed(1))),
                                         "sh0": torch.nn.Parameter(torch.randn(n, 1, 3, generator=torch.Generator().manual_seed(2))),
                                         "decoder_w": torch.nn.Parameter(torch.ones(4, 4))})
        opts = {k: torch.optim.Adam([params[k]], lr=1e-2) for k in params}
        return params, opts

    def train(params, opts, steps, weight):
        for _ in range(steps):
            loss = ((params["means"] ** 2).sum(-1) * weight).sum() + (params["sh0"].reshape(len(weight), -1).sum(-1) * weight).sum() \
                + params["decoder_w"].sum()
            for o in opts.values():
                o.zero_grad()
            loss.backward()
            for o in opts.values():
                o.step()

    w = torch.rand(n, generator=g)
    pa, oa = make()
    pb, ob = make()
    train(pa, oa, 3, w)
    train(pb, ob, 3, w)
    state = {"count": torch.arange(n, dtype=torch.float32), "scene_scale": torch.tensor(2.0)}
    perm = reorder_splats(pb, ob, state=state)
    assert torch.equal(perm, morton_order(pa["means"].detach())) and sorted(perm.tolist()) == list(range(n))
    assert torch.equal(state["count"], perm.float()) and float(state["scene_scale"]) == 2.0
    assert torch.equal(pb["decoder_w"], pa["decoder_w"])  # not a per-gaussian tensor: untouched, same object in its optimizer
    assert ob["decoder_w"].param_groups[0]["params"][0] is pb["decoder_w"]
    for k in ("means", "sh0"):
        assert torch.equal(pb[k].detach(), pa[k].detach()[perm]) and ob[k].param_groups[0]["params"][0] is pb[k]
        sa, sb = oa[k].state[pa[k]], ob[k].state[pb[k]]
        assert torch.equal(sb["exp_avg"], sa["exp_avg"][perm]) and torch.equal(sb["exp_avg_sq"], sa["exp_avg_sq"][perm])
        assert float(sb["step"]) == float(sa["step"])
    # ... and the training goes on identically
    train(pa, oa, 2, w)
    train(pb, ob, 2, w[perm])
    for k in ("means", "sh0"):
        assert torch.allclose(pb[k].detach(), pa[k].detach()[perm], rtol=0, atol=1e-7), k
    # an explicit permutation, plain dict, no optimizers
    d = {"means": torch.nn.Parameter(torch.arange(12.0).reshape(4, 3)), "opacities": torch.nn.Parameter(torch.arange(4.0))}
    reorder_splats(d, perm=torch.tensor([3, 1, 0, 2]))
    assert d["opacities"].tolist() == [3.0, 1.0, 0.0, 2.0] and d["means"][0].tolist() == [9.0, 10.0, 11.0]
