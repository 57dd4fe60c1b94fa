#!/usr/bin/env python
"""Generate the golden fixtures under tests/golden/ by IMPORTING THE REFERENCE (Python)
in the build container, and pin the CPU oracle against the reference at the same time.

Run (build container only; /root/reference does not exist on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it writes (data only -- inputs and the reference's outputs -- never reference source):
    garden_small.npz   derived fixture: 4k-point subsample of the reference's
                       assets/test_garden.npz crop + its 3 cameras + MATERIALISED seeded
                       scales / quats / opacities (the reference's
                       load_test_data draws them unseeded, gsplat/_helper.py:51-53)
    projection.npz     gsplat/cuda/_torch_impl.py:_fully_fused_projection outputs + autograd
                       gradients for {pinhole, ortho, fisheye} x compensations on/off
    sh.npz             _torch_impl._spherical_harmonics outputs + gradients, degree 0..4
    isect.npz          _torch_impl._isect_tiles / _isect_offset_encode on the recipe of the
                       reference's tests/test_basic.py:442-472 and on a garden-derived case
    quantize.npz       compression_simulation/ops.py fake_quantize_ste / STE on edge vectors
                       and random data (noise tensors captured)
    ../../gscodec_studio_amd/assets/garden_crop.npz
                       the [-2,2]^3 crop (111,785 points) + colours + cameras: the data half
                       of load_test_data, used by the bench workload generator

It also asserts that oracle/gs_oracle.c reproduces every one of those reference outputs
(bit-exact for integers and the quantizers, tight tolerances for fp32), i.e. running this
script IS the pinning of the oracle.  Compositing (R5) cannot be
pinned this way: the reference's _rasterize_to_pixels needs its CUDA extension + nerfacc.
"""
import math
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.dont_write_bytecode = True
sys.modules["_gridencoder"] = types.ModuleType("_gridencoder")  # CUDA-only hash-grid ext: stub
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

from gsplat.cuda import _torch_impl as T  # noqa: E402  (reference)
from gsplat.compression_simulation import ops as RQ  # noqa: E402  (reference)

from oracle import gs_oracle as O  # noqa: E402

torch.set_grad_enabled(True)


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")


def close(a, b, rtol, atol, what):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    err = np.abs(a - b) - (atol + rtol * np.abs(b))
    assert (err <= 0).all(), f"{what}: max violation {err.max():.3e} (max abs diff {np.abs(a-b).max():.3e})"
    print(f"  oracle == reference: {what} (max abs diff {np.abs(a - b).max():.2e})")


# ---------------------------------------------------------------------------
# garden fixtures
# ---------------------------------------------------------------------------
def make_garden():
    d = np.load(os.path.join(REF, "assets/test_garden.npz"))
    means = d["means3d"].astype(np.float32)
    colors = d["colors"]
    sel = ((means >= -2) & (means <= 2)).all(-1)  # load_test_data scene_crop (-2,-2,-2,2,2,2)
    means, colors = means[sel], colors[sel]
    assert len(means) == 111785, len(means)
    os.makedirs(os.path.join(REPO, "gscodec_studio_amd", "assets"), exist_ok=True)
    path = os.path.join(REPO, "gscodec_studio_amd", "assets", "garden_crop.npz")
    np.savez_compressed(path, means3d=means, colors=colors, viewmats=d["viewmats"].astype(np.float32),
                        Ks=d["Ks"].astype(np.float32), width=d["width"], height=d["height"])
    print(f"wrote {path}: {os.path.getsize(path) / 1024:.1f} KiB")

    g = torch.Generator().manual_seed(42)
    perm = torch.randperm(len(means), generator=g)[:4000].sort().values.numpy()
    m = means[perm]
    N = len(m)
    scales = (torch.rand((N, 3), generator=g) * 0.02).numpy()
    quats = torch.nn.functional.normalize(torch.randn((N, 4), generator=g), dim=-1).numpy()
    # also exercise un-normalised quaternions
    quats = quats * (0.5 + torch.rand((N, 1), generator=g).numpy())
    opacities = torch.rand((N,), generator=g).numpy()
    rgb = colors[perm].astype(np.float32) / 255.0
    # SH coefficients are derived in the tests (tests/util.py:garden_sh) to keep the fixture small
    fx = dict(means=m, scales=scales.astype(np.float32), quats=quats.astype(np.float32),
              opacities=opacities.astype(np.float32), rgb=rgb, viewmats=d["viewmats"].astype(np.float32),
              Ks=d["Ks"].astype(np.float32), width=np.int64(d["width"]), height=np.int64(d["height"]))
    save("garden_small.npz", **fx)
    return fx


# ---------------------------------------------------------------------------
# projection
# ---------------------------------------------------------------------------
def make_projection(fx):
    out = {}
    N = 1200
    means = torch.tensor(fx["means"][:N])
    quats = torch.tensor(fx["quats"][:N])
    scales = torch.tensor(fx["scales"][:N] * 3.0)  # a bit larger so that radii are interesting
    viewmats = torch.tensor(fx["viewmats"])
    Ks = torch.tensor(fx["Ks"])
    W, H = int(fx["width"]), int(fx["height"])
    out.update(means=means.numpy(), quats=quats.numpy(), scales=scales.numpy(), viewmats=viewmats.numpy(),
               Ks=Ks.numpy(), width=np.int64(W), height=np.int64(H))
    g = torch.Generator().manual_seed(7)
    C = viewmats.shape[0]
    v_means2d = torch.randn((C, N, 2), generator=g)
    v_depths = torch.randn((C, N), generator=g)
    v_conics = torch.randn((C, N, 3), generator=g)
    v_comp = torch.randn((C, N), generator=g)
    out.update(v_means2d=v_means2d.numpy(), v_depths=v_depths.numpy(), v_conics=v_conics.numpy(), v_comp=v_comp.numpy())
    for model in ["pinhole", "ortho", "fisheye"]:
        for comp in [False, True]:
            tag = f"{model}_{int(comp)}"
            mm, qq, ss, vv = (t.clone().requires_grad_(True) for t in (means, quats, scales, viewmats))
            covars, _ = T._quat_scale_to_covar_preci(qq, ss, triu=False)
            radii, means2d, depths, conics, comps = T._fully_fused_projection(
                mm, covars, vv, Ks, W, H, calc_compensations=comp, camera_model=model)
            valid = (radii > 0)
            loss = (means2d * v_means2d * valid[..., None]).sum() + (depths * v_depths * valid).sum() + \
                   (conics * v_conics * valid[..., None]).sum()
            if comp:
                loss = loss + (comps * v_comp * valid).sum()
            g_m, g_q, g_s, g_v = torch.autograd.grad(loss, (mm, qq, ss, vv))
            out[f"{tag}_radii"] = radii.numpy().astype(np.int32)
            out[f"{tag}_means2d"] = means2d.detach().numpy()
            out[f"{tag}_depths"] = depths.detach().numpy()
            out[f"{tag}_conics"] = conics.detach().numpy()
            if comp:
                out[f"{tag}_comp"] = comps.detach().numpy()
            out[f"{tag}_v_means"] = g_m.numpy(); out[f"{tag}_v_quats"] = g_q.numpy()
            out[f"{tag}_v_scales"] = g_s.numpy(); out[f"{tag}_v_viewmats"] = g_v.numpy()

            # ---- pin the oracle
            o_r, o_m2, o_d, o_c, o_cp = O.projection_fwd(means.numpy(), None, quats.numpy(), scales.numpy(),
                                                         viewmats.numpy(), Ks.numpy(), W, H,
                                                         calc_compensations=comp, camera_model=model)
            r_ref = out[f"{tag}_radii"]
            assert (np.abs(o_r - r_ref) <= 1).all(), f"{tag}: radii differ by more than 1"
            frac_exact = (o_r == r_ref).mean()
            both = (o_r > 0) & (r_ref > 0)
            print(f"[{tag}] visible {both.sum()} / {both.size}, radii exact {frac_exact*100:.3f}%")
            close(o_m2[both], out[f"{tag}_means2d"][both], 1e-5, 1e-4, f"{tag} means2d")
            close(o_d[both], out[f"{tag}_depths"][both], 1e-6, 1e-6, f"{tag} depths")
            close(o_c[both], out[f"{tag}_conics"][both], 2e-4, 1e-5, f"{tag} conics")
            if comp:
                close(o_cp[both], out[f"{tag}_comp"][both], 1e-4, 5e-4, f"{tag} compensations")  # reference test: atol 1e-3
            # bwd with the reference's own radii / conics so that both sides see the same visibility
            vm = (v_means2d * valid[..., None]).numpy(); vd = (v_depths * valid).numpy()
            vc = (v_conics * valid[..., None]).numpy(); vcp = (v_comp * valid).numpy() if comp else None
            b_m, _, b_q, b_s, b_v = O.projection_bwd(
                means.numpy(), None, quats.numpy(), scales.numpy(), viewmats.numpy(), Ks.numpy(), W, H, 0.3, model,
                r_ref, out[f"{tag}_conics"], out.get(f"{tag}_comp"), vm, vd, vc, vcp)
            close(b_m, g_m.numpy(), 2e-3, 2e-3 * np.abs(g_m.numpy()).max(), f"{tag} v_means")
            close(b_q, g_q.numpy(), 2e-3, 2e-3 * np.abs(g_q.numpy()).max(), f"{tag} v_quats")
            close(b_s, g_s.numpy(), 2e-3, 2e-3 * np.abs(g_s.numpy()).max(), f"{tag} v_scales")
            close(b_v, g_v.numpy(), 2e-3, 2e-3 * np.abs(g_v.numpy()).max(), f"{tag} v_viewmats")
    save("projection.npz", **out)


# ---------------------------------------------------------------------------
# spherical harmonics (recipe of tests/test_basic.py:579-607)
# ---------------------------------------------------------------------------
def make_sh():
    g = torch.Generator().manual_seed(42)
    N, K = 400, 25
    coeffs = torch.randn((N, K, 3), generator=g)
    dirs = torch.randn((N, 3), generator=g)
    v_colors = torch.randn((N, 3), generator=g)
    out = dict(coeffs=coeffs.numpy(), dirs=dirs.numpy(), v_colors=v_colors.numpy())
    for deg in range(5):
        cc, dd = coeffs.clone().requires_grad_(True), dirs.clone().requires_grad_(True)
        colors = T._spherical_harmonics(deg, dd, cc)
        g_c, g_d = torch.autograd.grad((colors * v_colors).sum(), (cc, dd), allow_unused=True)
        if g_d is None:
            g_d = torch.zeros_like(dirs)
        out[f"deg{deg}_colors"] = colors.detach().numpy()
        out[f"deg{deg}_v_coeffs"] = g_c.numpy()
        out[f"deg{deg}_v_dirs"] = g_d.numpy()
        o_col = O.sh_fwd(deg, dirs.numpy(), coeffs.numpy())
        o_vc, o_vd = O.sh_bwd(deg, dirs.numpy(), coeffs.numpy(), v_colors.numpy())
        close(o_col, out[f"deg{deg}_colors"], 1e-5, 1e-5, f"sh deg{deg} colors")
        close(o_vc, out[f"deg{deg}_v_coeffs"], 1e-5, 1e-5, f"sh deg{deg} v_coeffs")
        close(o_vd, out[f"deg{deg}_v_dirs"], 1e-4, 1e-4, f"sh deg{deg} v_dirs")
    save("sh.npz", **out)


# ---------------------------------------------------------------------------
# tile intersection (recipe of tests/test_basic.py:442-472) -- bit exact
# ---------------------------------------------------------------------------
def ref_isect(means2d, radii, depths, tile_size, tw, th):
    C = means2d.shape[0]
    tpg, ids, flat = T._isect_tiles(means2d, radii, depths, tile_size, tw, th, sort=False)
    # the reference's torch.sort is not stable; CUB's radix sort is.  Emission order is
    # ascending flatten id, so "stable" == ties broken by position in the unsorted list.
    order = np.lexsort((np.arange(len(ids)), ids.numpy()))
    ids_s, flat_s = ids.numpy()[order], flat.numpy()[order]
    offs = T._isect_offset_encode(torch.from_numpy(ids_s), C, tw, th)
    return tpg.numpy().astype(np.int32), ids.numpy(), flat.numpy().astype(np.int32), ids_s, flat_s.astype(np.int32), offs.numpy().astype(np.int32)


def make_isect(fx):
    out = {}
    # (a) the reference test's recipe
    torch.manual_seed(42)
    C, N, width, height, ts = 3, 1000, 40, 60, 16
    means2d = torch.randn(C, N, 2) * width
    radii = torch.randint(0, width, (C, N), dtype=torch.int32)
    depths = torch.rand(C, N)
    # force exact depth ties inside shared tiles to pin stability
    depths[0, 100:140] = depths[0, 100]
    means2d[0, 100:140] = torch.tensor([20.0, 30.0])
    radii[0, 100:140] = 9
    tw, th = math.ceil(width / ts), math.ceil(height / ts)
    tpg, ids_u, flat_u, ids_s, flat_s, offs = ref_isect(means2d, radii, depths, ts, tw, th)
    out.update(a_means2d=means2d.numpy(), a_radii=radii.numpy(), a_depths=depths.numpy(), a_tile_size=np.int64(ts),
               a_tile_width=np.int64(tw), a_tile_height=np.int64(th), a_tiles_per_gauss=tpg, a_isect_ids_unsorted=ids_u,
               a_flatten_ids_unsorted=flat_u, a_isect_ids=ids_s, a_flatten_ids=flat_s, a_isect_offsets=offs)
    o_tpg, o_ids, o_flat = O.isect_tiles(means2d.numpy(), radii.numpy(), depths.numpy(), ts, tw, th)
    assert (o_tpg == tpg).all() and (o_ids == ids_s).all() and (o_flat == flat_s).all()
    assert (O.isect_offset_encode(o_ids, C, tw, th) == offs).all()
    _, o_ids_u, o_flat_u = O.isect_tiles(means2d.numpy(), radii.numpy(), depths.numpy(), ts, tw, th, sort=False)
    assert (o_ids_u == ids_u).all() and (o_flat_u == flat_u).all()
    print(f"  oracle == reference (bit exact): isect recipe (a), n_isects = {len(ids_s)}")

    # (b) garden-derived: project 1500 gaussians with the reference, 2 cameras, 648x420
    Nb = 1500
    covars, _ = T._quat_scale_to_covar_preci(torch.tensor(fx["quats"][:Nb]), torch.tensor(fx["scales"][:Nb] * 4.0))
    W, H = int(fx["width"]), int(fx["height"])
    radii, means2d, depths, _, _ = T._fully_fused_projection(
        torch.tensor(fx["means"][:Nb]), covars, torch.tensor(fx["viewmats"][:2]), torch.tensor(fx["Ks"][:2]), W, H)
    means2d, depths = means2d.detach(), depths.detach()
    tw, th = math.ceil(W / 16), math.ceil(H / 16)
    tpg, ids_u, flat_u, ids_s, flat_s, offs = ref_isect(means2d, radii, depths, 16, tw, th)
    out.update(b_means2d=means2d.numpy(), b_radii=radii.numpy().astype(np.int32), b_depths=depths.numpy(),
               b_tile_size=np.int64(16), b_tile_width=np.int64(tw), b_tile_height=np.int64(th), b_tiles_per_gauss=tpg,
               b_isect_ids=ids_s, b_flatten_ids=flat_s, b_isect_offsets=offs)
    o_tpg, o_ids, o_flat = O.isect_tiles(means2d.numpy(), radii.numpy().astype(np.int32), depths.numpy(), 16, tw, th)
    assert (o_tpg == tpg).all() and (o_ids == ids_s).all() and (o_flat == flat_s).all()
    assert (O.isect_offset_encode(o_ids, 2, tw, th) == offs).all()
    print(f"  oracle == reference (bit exact): isect garden case (b), n_isects = {len(ids_s)}")
    save("isect.npz", **out)


# ---------------------------------------------------------------------------
# quantizers -- bit exact
# ---------------------------------------------------------------------------
def make_quant():
    out = {}
    g = torch.Generator().manual_seed(3)
    bounds = {"scales": (-10, 2), "quats": (-1, 1), "opacities": (-15, 15), "sh0": (-2, 4), "stg_opacities": (-7, 7),
              "stg_colors": (-7.5, 7.5), "features": (-10, 10)}
    edge = torch.tensor([-3, -0.4, 0, 0.30001, 0.9, 2.5], dtype=torch.float32)
    for name, (lo, hi) in bounds.items():
        n = 1500
        x = (torch.rand(n, generator=g) * 1.4 - 0.2) * (hi - lo) + lo  # ~14% out of range on each side
        # exact grid points and exact .5 ties on the 8-bit grid
        q = (hi - lo) / 255.0
        ties = torch.tensor([lo + (k + 0.5) * q for k in range(0, 255, 5)], dtype=torch.float32)
        grid = torch.tensor([lo + k * q for k in range(0, 256, 5)], dtype=torch.float32)
        x = torch.cat([x, ties, grid, edge, torch.tensor([lo, hi], dtype=torch.float32)])
        out[f"{name}_x"] = x.numpy()
        for bits in (8, 4):
            # round / STE (mutates its input: pass a clone, record the clamped parameter too)
            xin = x.clone().requires_grad_(False)
            leaf = xin.clone().requires_grad_(True)
            work = leaf.detach()  # shares storage with leaf, like param.data
            res = RQ.fake_quantize_ste(work, lo, hi, bits, "round")
            out[f"{name}_round{bits}_out"] = res["output_value"].numpy()
            out[f"{name}_round{bits}_x_after"] = work.numpy().copy()
            o_x, o_out = O.quant_round_fwd(x.numpy(), lo, hi, bits)
            assert (o_out.view(np.uint32) == out[f"{name}_round{bits}_out"].view(np.uint32)).all(), name
            assert (o_x.view(np.uint32) == out[f"{name}_round{bits}_x_after"].view(np.uint32)).all(), name
            # noise: capture the noise tensor by replaying the generator state
            torch.manual_seed(1234 + bits)
            state = torch.get_rng_state()
            leaf = x.clone().requires_grad_(True)
            res = RQ.fake_quantize_ste(leaf, lo, hi, bits, "noise")
            torch.set_rng_state(state)
            noise = torch.empty_like(x).uniform_(-0.5, 0.5)
            v_out = torch.randn(x.shape, generator=g)
            (v_x,) = torch.autograd.grad((res["output_value"] * v_out).sum(), leaf)
            out[f"{name}_noise{bits}_noise"] = noise.numpy()
            out[f"{name}_noise{bits}_out"] = res["output_value"].detach().numpy()
            out[f"{name}_noise{bits}_v_out"] = v_out.numpy()
            out[f"{name}_noise{bits}_v_x"] = v_x.numpy()
            out[f"{name}_noise{bits}_q_step"] = np.float64(res["q_step"])
            o = O.quant_noise_fwd(x.numpy(), noise.numpy(), lo, hi, res["q_step"])
            assert (o.view(np.uint32) == out[f"{name}_noise{bits}_out"].view(np.uint32)).all(), name
            o = O.quant_noise_bwd(x.numpy(), v_out.numpy(), lo, hi)
            assert (o.view(np.uint32) == v_x.numpy().view(np.uint32)).all(), name
        out[f"{name}_bounds"] = np.array([lo, hi], np.float64)
    print("  oracle == reference (bit exact): quantizers, all bound sets, 8 and 4 bits")
    save("quantize.npz", **out)


if __name__ == "__main__":
    O.build()
    fx = make_garden()
    make_projection(fx)
    make_sh()
    make_isect(fx)
    make_quant()
    print("all golden fixtures written; oracle pinned against the reference")
