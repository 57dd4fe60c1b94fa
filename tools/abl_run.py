"""Debug: run the bench workload once with an ablation build and print its counters (GS_ABL=9)."""
import ctypes, os, sys, subprocess
lib = os.environ["GSPLAT_HIP_LIB"]
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa
sys.argv = ["bench.py", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
bench.main()
from gscodec_studio_amd import _backend as B
L = ctypes.CDLL(lib)
out = (ctypes.c_ulonglong * 16)()
L.gs_debug_abl_stats(out)
n = out[4] or 1
print("batches", out[4], "visits", out[0], "passes", out[1], "reduces", out[2], "valid lanes", out[3], "empty passes", out[5], "alpha-empty passes", out[6], "(all summed over 4 bwd launches)")
import numpy as np
w = (ctypes.c_ulonglong * (65536 * 4))()
L.gs_debug_abl_waves(w)
w = np.frombuffer(w, dtype=np.uint64).reshape(-1, 4)
w = w[w[:, 1] > 0]
t0 = w[:, 0].astype(np.int64); t1 = w[:, 1].astype(np.int64); ln = (w[:, 2] >> np.uint64(32)).astype(np.int64); ev = (w[:, 2] & np.uint64(0xffffffff)).astype(np.int64)
cyc = (w[:, 3] >> np.uint64(24)).astype(np.int64); bar = ((w[:, 3] & np.uint64(0xffffff)) << np.uint64(8)).astype(np.int64)
if len(w):
    k0 = t0.min()
    print("fwd waves", len(w), "kernel span us", (t1.max() - k0) / 100.0, "sum wave us", (t1 - t0).sum() / 100.0)
    o = np.argsort(-(t1 - t0))[:8]
    for i in o:
        print("  wave", i, "start us", (t0[i] - k0) / 100.0, "dur us", (t1[i] - t0[i]) / 100.0, "cycles", cyc[i], "barrier-wait cycles", bar[i], "list", ln[i], "evals", ev[i])
    o = np.argsort(-t1)[:5]
    for i in o:
        print("  last-ending wave", i, "start us", (t0[i] - k0) / 100.0, "end us", (t1[i] - k0) / 100.0, "list", ln[i], "evals", ev[i])
    late = np.sort(t0 - k0)
    print("start time percentiles us", [float(np.percentile(late, q)) / 100.0 for q in (50, 90, 99, 100)])
    print("total cycles", cyc.sum(), "barrier cycles", bar.sum(), "evals", ev.sum())
bw = (ctypes.c_ulonglong * (65536 * 2))()
L.gs_debug_abl_bwaves(bw)
bw = np.frombuffer(bw, dtype=np.uint64).reshape(-1, 2)
bw = bw[bw[:, 1] > 0].astype(np.int64)
if len(bw):
    bw = bw[bw[:, 0] >= bw[:, 1].max() - 1000000]  # the last launch only (10 ms window; the clock ticks at 100 MHz)
    k0 = bw[:, 0].min()
    st, en, du = (bw[:, 0] - k0) / 100.0, (bw[:, 1] - k0) / 100.0, (bw[:, 1] - bw[:, 0]) / 100.0
    print("bwd items", len(bw), "kernel span us", en.max(), "sum item us", du.sum(), "-> avg concurrency", du.sum() / en.max())
    print("  start percentiles us", [float(np.percentile(st, q)) for q in (50, 90, 99, 100)])
    print("  end   percentiles us", [float(np.percentile(en, q)) for q in (50, 90, 99, 100)])
    print("  duration percentiles us", [float(np.percentile(du, q)) for q in (10, 50, 90, 99, 100)])
