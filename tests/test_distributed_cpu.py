"""CPU, world_size 2, gloo: the multi-GPU layer (collectives of the reference's
tests/_test_distributed.py:13-107, the gaussian-sharded exchange and the splat-gradient
reduction of the camera-sharded data-parallel path)."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, fn_name):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        globals()[fn_name](rank, world)
        dist.barrier()
    finally:  # (no barrier here: after an exception on one rank it would never return)
        dist.destroy_process_group()


def _run(fn_name, world=2):
    mp.spawn(_worker, args=(world, _free_port(), fn_name), nprocs=world, join=True)


# --------------------------------------------------------------------------- collectives
def _collectives(rank, world):
    from gscodec_studio_amd import distributed as D

    dev = torch.device("cpu")
    # all_gather_int32 (reference _test_distributed.py:13-28)
    assert D.all_gather_int32(world, rank + 10, device=dev) == [10, 11]
    got = D.all_gather_int32(world, torch.tensor(rank + 10, dtype=torch.int32))
    assert [int(t) for t in got] == [10, 11]
    # all_to_all_int32 (30-47)
    vals = [rank * 2 + 0, rank * 2 + 1]
    assert D.all_to_all_int32(world, vals, device=dev) == [0 + rank, 2 + rank]
    # all_gather_tensor_list (49-75), with gradients
    a = torch.full((2, 3), float(rank), requires_grad=True)
    b = torch.full((2,), float(rank) + 0.5)
    ga, gb = D.all_gather_tensor_list(world, [a, b])
    assert ga.shape == (4, 3) and gb.shape == (4,)
    assert torch.equal(ga[:2], torch.zeros(2, 3)) and torch.equal(ga[2:], torch.ones(2, 3))
    (ga * (rank + 1)).sum().backward()
    assert torch.allclose(a.grad, torch.full((2, 3), 3.0))  # (1 + 2): every rank used my slab
    # all_to_all_tensor_list (77-107): the docstring example of the reference
    if rank == 0:
        tl, splits = [torch.tensor([1., 2., 3.], requires_grad=True), torch.tensor([4., 5., 6.])], [2, 1]
    else:
        tl, splits = [torch.tensor([7., 8.], requires_grad=True), torch.tensor([9., 10.])], [1, 1]
    x, y = D.all_to_all_tensor_list(world, tl, splits)
    if rank == 0:
        assert x.tolist() == [1, 2, 7] and y.tolist() == [4, 5, 9]
    else:
        assert x.tolist() == [3, 8] and y.tolist() == [6, 10]
    (x * (10.0 ** rank)).sum().backward()
    if rank == 0:
        assert tl[0].grad.tolist() == [1, 1, 10]
    else:
        assert tl[0].grad.tolist() == [1, 10]


def test_collectives_world2():
    _run("_collectives")


# --------------------------------------------------------------------------- gaussian-sharded exchange
def _exchange(rank, world):
    from gscodec_studio_amd import distributed as D

    # 2 ranks, N_world = [3, 5] gaussians, 1 camera each -> C_total = 2
    N_world, C_world = [3, 5], [1, 1]
    N = N_world[rank]
    C = 2
    off = sum(N_world[:rank])
    # value of element (c, n_global) = 100 * c + n_global  -> easy to verify after the exchange
    base = (100 * torch.arange(C)[:, None] + (off + torch.arange(N))[None, :]).float()
    radii = base.to(torch.int32) + 1
    means2d = torch.stack([base, -base], -1)
    depths, conics, opac, colors = base + 0.25, base[..., None].repeat(1, 1, 3), base + 0.5, base[..., None].repeat(1, 1, 3) * 2
    out = D.exchange_projected(rank, world, N, N_world, C_world, False, radii, means2d, depths, conics, opac, colors, None, None)
    C_local, r2, m2, d2, c2, o2, col2, cam, gau = out
    assert C_local == 1 and cam is None and gau is None
    expect = (100 * rank + torch.arange(sum(N_world))).float()[None]
    assert torch.equal(r2, expect.to(torch.int32) + 1)
    assert torch.equal(m2[..., 0], expect) and torch.equal(d2, expect + 0.25) and torch.equal(col2[..., 2], expect * 2)
    # packed: rank r sees gaussians visible in both cameras, ids local -> global
    cam_ids = torch.tensor([0, 0, 1], dtype=torch.int64)
    gau_ids = torch.tensor([0, 2, 1], dtype=torch.int64)
    vals = (1000 * rank + 100 * cam_ids + gau_ids).float()
    out = D.exchange_projected(rank, world, N, N_world, C_world, True, (vals + 1).to(torch.int32), torch.stack([vals, vals], -1),
                               vals, vals[:, None].repeat(1, 3), vals, vals[:, None].repeat(1, 3), cam_ids, gau_ids)
    C_local, r2, m2, d2, c2, o2, col2, cam, gau = out
    assert (cam == 0).all()  # local camera index on every rank
    if rank == 0:   # receives camera-0 rows of rank 0 (2 rows) and of rank 1 (2 rows)
        assert d2.tolist() == [0., 2., 1000., 1002.] and gau.tolist() == [0, 2, 3, 5]
    else:           # camera-1 rows
        assert d2.tolist() == [101., 1101.] and gau.tolist() == [1, 4]


def test_gaussian_sharded_exchange_world2():
    _run("_exchange")


# --------------------------------------------------------------------------- gradient reduction
def _grad_reduce(rank, world):
    from gscodec_studio_amd import distributed as D

    torch.manual_seed(0)
    params = {"means": torch.nn.Parameter(torch.randn(7, 3)), "sh": torch.nn.Parameter(torch.randn(7, 4, 3)),
              "frozen": torch.nn.Parameter(torch.randn(2), requires_grad=False), "nograd": torch.nn.Parameter(torch.randn(5))}
    g = {k: torch.full_like(v, float(rank + 1)) for k, v in params.items()}
    params["means"].grad = g["means"].clone()
    params["sh"].grad = g["sh"].clone()
    D.all_reduce_splat_grads(params, average=False)
    assert torch.equal(params["means"].grad, torch.full((7, 3), 3.0))
    assert torch.equal(params["sh"].grad, torch.full((7, 4, 3), 3.0))
    assert torch.equal(params["nograd"].grad, torch.zeros(5))  # missing grads count as zeros on every rank
    assert params["frozen"].grad is None
    params["means"].grad = g["means"].clone()
    D.all_reduce_splat_grads([params["means"]], average=True, algorithm="all_reduce")
    assert torch.allclose(params["means"].grad, torch.full((7, 3), 1.5))
    # the copy-free per-tensor path (default on RCCL; on gloo it falls back to in-place all_reduce per tensor)
    params["means"].grad = g["means"].clone()
    params["sh"].grad = g["sh"].clone().transpose(1, 2).contiguous().transpose(1, 2)  # non-contiguous gradient
    params["nograd"].grad = None
    D.all_reduce_splat_grads(params, average=True, algorithm="direct")
    assert torch.allclose(params["means"].grad, torch.full((7, 3), 1.5))
    assert torch.allclose(params["sh"].grad, torch.full((7, 4, 3), 1.5)) and params["sh"].grad.is_contiguous()
    assert torch.equal(params["nograd"].grad, torch.zeros(5))
    # camera-sharded equivalence on a toy differentiable "renderer": sum over ALL cameras of f(c, theta)
    theta = torch.nn.Parameter(torch.arange(6, dtype=torch.float32).reshape(3, 2))
    cams = torch.arange(1, 5, dtype=torch.float32)  # 4 cameras
    mine = D.shard_cameras(4, rank, world)
    loss = sum(((theta * cams[c]) ** 2).sum() for c in mine)
    loss.backward()
    D.all_reduce_splat_grads([theta], average=False)
    full = torch.arange(6, dtype=torch.float32).reshape(3, 2).requires_grad_(True)
    sum(((full * c) ** 2).sum() for c in cams).backward()
    assert torch.allclose(theta.grad, full.grad)


def test_splat_grad_reduction_world2():
    _run("_grad_reduce")


def _grad_reduce_span(rank, world):
    """The one-span route of algorithm="direct" (forced at toy size): the collective's length depends on the parameters'
    sizes only, so a rank whose gradients are missing / sparse / separately allocated joins the same collective as a rank
    whose gradients lie in one buffer; a SUBSET of the parameters leaves the others' gradients alone."""
    os.environ["GS_DP_RS_AG_MIN_BYTES"] = "64"
    from gscodec_studio_amd import distributed as D

    assert D._DIRECT_RS_AG_MIN_BYTES == 64
    assert D._span_length([7 * 3, 100, 64, 1], 2) == 64 + 128 + 64 + 64 and D._span_length([5], 3) == 66
    torch.manual_seed(0)
    shapes = {"means": (7, 3), "sh": (50, 2), "quats": (16, 4), "opac": (1,)}
    params = {k: torch.nn.Parameter(torch.randn(*s)) for k, s in shapes.items()}
    if rank == 0:  # one buffer, carved like _wrapper.GradPrefill (256-byte aligned pieces)
        buf = torch.zeros(D._span_length([p.numel() for p in params.values()], world) + 64)
        off = 0
        for p in params.values():
            p.grad = buf[off:off + p.numel()].view(p.shape)
            p.grad.fill_(1.0)
            off += (p.numel() + 63) // 64 * 64
    else:  # separate tensors, one missing
        for k, p in params.items():
            p.grad = None if k == "quats" else torch.full_like(p, 2.0)
    D.all_reduce_splat_grads(params, average=False, algorithm="direct")
    for k, p in params.items():
        want = 1.0 if (k == "quats") else 3.0
        assert torch.equal(p.grad, torch.full(shapes[k], want)), (k, p.grad.flatten()[:4])
    # a subset, averaged; then the rest, summed: the rest must not have been touched by the first call
    for p in params.values():
        p.grad.fill_(float(rank + 1))
    D.all_reduce_splat_grads([params["means"], params["quats"]], average=True, algorithm="direct")
    assert torch.equal(params["means"].grad, torch.full(shapes["means"], 1.5))
    assert torch.equal(params["sh"].grad, torch.full(shapes["sh"], float(rank + 1)))
    D.all_reduce_splat_grads([params["sh"], params["opac"]], average=False, algorithm="direct")
    assert torch.equal(params["sh"].grad, torch.full(shapes["sh"], 3.0)) and torch.equal(params["opac"].grad, torch.full((1,), 3.0))
    assert torch.equal(params["quats"].grad, torch.full(shapes["quats"], 1.5))


def test_splat_grad_reduction_one_span_world2():
    _run("_grad_reduce_span")


def _grad_reduce_span_trainer_order(rank, world):
    """Round-4 advisor finding: the reference trainer's dict order (means, scales, quats, ...: examples/simple_trainer.py) is
    NOT the order rasterization() carves the gradients in (means, quats, scales, ...).  Rank 0 reduces IN PLACE on the carved
    buffer, rank 1 stages (separately allocated gradients): both must use one layout, or quats [N,4] and scales [N,3] are added
    at each other's offsets.  Random values, so any mix-up shows."""
    os.environ["GS_DP_RS_AG_MIN_BYTES"] = "64"
    from gscodec_studio_amd import distributed as D

    N = 37
    shapes = {"means": (N, 3), "scales": (N, 3), "quats": (N, 4), "opacities": (N,), "sh0": (N, 1, 3), "shN": (N, 15, 3)}
    carve_order = ["means", "quats", "scales", "opacities", "sh0", "shN"]
    assert [list(shapes)[i] for i in D._canonical_order(list(shapes))] == carve_order
    assert D._canonical_order([None, None, None]) == [0, 1, 2]
    assert [["x", "shN", "means"][i] for i in D._canonical_order(["x", "shN", "means"])] == ["means", "shN", "x"]
    g = torch.Generator().manual_seed(5)
    vals = {k: torch.randn(world, *shp, generator=g) for k, shp in shapes.items()}
    params = {k: torch.nn.Parameter(torch.zeros(shp)) for k, shp in shapes.items()}
    length = D._span_length([params[k].numel() for k in carve_order], world)
    if rank == 0:
        buf = torch.full((length + 200,), -7.0)
        off = 72  # (the compositing gradient rows lie in front)
        for k in carve_order:
            p = params[k]
            p.grad = buf[off:off + p.numel()].view(p.shape)
            p.grad.copy_(vals[k][rank])
            off += (p.numel() + 63) // 64 * 64
        assert D._one_span([params[k] for k in carve_order], length) is not None     # taken in place ...
        assert D._one_span(list(params.values()), length) is None                    # ... but never in the dict's order
    else:
        for k, p in params.items():
            p.grad = vals[k][rank].clone()
    D.all_reduce_splat_grads(params, average=False, algorithm="direct")
    for k, p in params.items():
        assert torch.allclose(p.grad, vals[k].sum(0), atol=1e-6), k
    if rank == 0:
        assert bool((buf[:72] == -7.0).all())
        assert params["quats"].grad.untyped_storage().data_ptr() == buf.untyped_storage().data_ptr()  # (still in place)
    # as a LIST in the dict's order (no names: the list order is the layout), rank 0's carved buffer no longer matches -> staged
    for k, p in params.items():
        p.grad.copy_(vals[k][rank])
    D.all_reduce_splat_grads(list(params.values()), average=True, algorithm="direct")
    for k, p in params.items():
        assert torch.allclose(p.grad, vals[k].mean(0), atol=1e-6), k


def test_splat_grad_reduction_trainer_dict_order_world2():
    _run("_grad_reduce_span_trainer_order")


# --------------------------------------------------------------------------- sparse gradient reduction (camera-sharded)
def _sparse_grad_reduce(rank, world):
    """plan_sparse_grad_exchange + all_reduce_splat_grads(plan=...) equals the dense sum: ragged N (not a multiple of the
    world size), ranks seeing different subsets (overlapping, disjoint, one splat seen by nobody, one owner block that
    nobody sees anything in), several parameters of different widths, average on and off."""
    from gscodec_studio_amd import distributed as D

    for N, seed in ((11, 0), (64, 1), (5, 2)):
        g = torch.Generator().manual_seed(seed)
        vis_all = torch.rand(world, 2, N, generator=g) < 0.45           # [rank, camera, splat]
        vis_all[:, :, 0] = False                                        # nobody sees splat 0
        if N == 64:
            vis_all[:, :, 32:] = False                                  # nobody sees anything in the second owner block
        radii = vis_all[rank].to(torch.int32) * 3
        plan = D.plan_sparse_grad_exchange(radii, world)
        assert plan is not None and plan.block == -(-N // world)
        shapes = {"means": (N, 3), "opacities": (N,), "sh": (N, 4, 3)}
        full = {k: torch.randn(world, *shp, generator=g) for k, shp in shapes.items()}   # what every rank WOULD have
        seen = vis_all.any(1)                                                          # [rank, N]
        params = {}
        for k, shp in shapes.items():
            p = torch.nn.Parameter(torch.zeros(shp))
            m = seen[rank].reshape((N,) + (1,) * (len(shp) - 1))
            p.grad = full[k][rank] * m                                                  # zero where this rank saw nothing
            params[k] = p
        expect = {k: sum(full[k][r] * seen[r].reshape((N,) + (1,) * (len(shapes[k]) - 1)) for r in range(world)) for k in shapes}
        D.WIRE["bytes"] = 0
        D.all_reduce_splat_grads(params, average=False, plan=plan)
        for k in shapes:
            assert torch.allclose(params[k].grad, expect[k], atol=1e-6), (N, k)
        rows, urows = plan.counts()
        assert rows[rank] == [int(seen[rank][o * plan.block:(o + 1) * plan.block].sum()) for o in range(world)]
        assert urows == [int(seen.any(0)[o * plan.block:(o + 1) * plan.block].sum()) for o in range(world)]
        # wire: rows sent to the OTHER owner + the padded union block gathered to the other rank (16 floats per row + the
        # splat index every wire row carries in its first column)
        assert D.WIRE["bytes"] == 4 * 17 * (rows[rank][1 - rank] + max(urows) * (world - 1))
        # averaged form
        for k in shapes:
            params[k].grad = full[k][rank] * seen[rank].reshape((N,) + (1,) * (len(shapes[k]) - 1))
        D.all_reduce_splat_grads(params, average=True, plan=plan)
        for k in shapes:
            assert torch.allclose(params[k].grad, expect[k] / world, atol=1e-6), (N, k)


def test_sparse_splat_grad_reduction_world2():
    _run("_sparse_grad_reduce")
