import os, sys, subprocess, numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
sys.path.insert(0, os.path.join(R, "tests"))
if len(sys.argv) > 1:
    import torch
    from tests.test_gpu_rasterization import _inputs, T, N
    from gscodec_studio_amd import rasterization
    d = _inputs(n=3000, cams=2, sh_degree=None)
    rc, ra, meta = rasterization(T(d["means"]), T(d["quats"]), T(d["scales"]), T(d["opacities"]), T(d["colors"]), T(d["viewmats"]), T(d["Ks"]), d["W"], d["H"], packed=False)
    np.savez(sys.argv[1], rc=N(rc), ra=N(ra), offs=N(meta["isect_offsets"]))
else:
    env = dict(os.environ)
    subprocess.check_call([sys.executable, __file__, "/tmp/new.npz"], env=env)
    env["GS_RASTER_FWD"] = "wave"
    subprocess.check_call([sys.executable, __file__, "/tmp/old.npz"], env=env)
    a, b = np.load("/tmp/new.npz"), np.load("/tmp/old.npz")
    bad = np.abs(a["ra"] - b["ra"])[..., 0] > 1e-4
    print("shape", bad.shape, "bad frac", bad.mean())
    C, H, W = bad.shape
    offs = a["offs"]
    print("tiles", offs.shape)
    for c in range(C):
        ys, xs = np.nonzero(bad[c])
        if len(ys) == 0:
            continue
        tiles = sorted(set(zip((ys // 16).tolist(), (xs // 16).tolist())))
        print("cam", c, "bad tiles", len(tiles), "of", offs.shape[1] * offs.shape[2])
        flat = offs.reshape(-1)
        for (ty, tx) in tiles[:12]:
            lin = (c * offs.shape[1] + ty) * offs.shape[2] + tx
            rs = flat[lin]; re = flat[lin + 1] if lin + 1 < len(flat) else -1
            sub = bad[c, ty * 16:(ty + 1) * 16, tx * 16:(tx + 1) * 16]
            qs = [int(sub[:8, :8].sum()), int(sub[:8, 8:].sum()), int(sub[8:, :8].sum()), int(sub[8:, 8:].sum())]
            print("  tile", ty, tx, "range", rs, re, "len", re - rs, "rs%64", rs % 64, "bad per quadrant", qs)
