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
out = (ctypes.c_ulonglong * 8)()
L.gs_debug_abl_stats(out)
n = out[4] or 1
print("batches", out[4], "visits", out[0], "passes", out[1], "reduces", out[2], "valid lanes", out[3], "empty passes", out[5], "alpha-empty passes", out[6], "(all summed over 4 bwd launches)")
