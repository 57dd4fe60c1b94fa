"""bench.py started the way the DRIVER starts it -- ``python bench.py --gpus N ...`` with no launcher around it -- must bring up
its own N ranks (the reference's harness does the same: profiling/main.py:370 -> gsplat/distributed.py:304-360), leave rank 0's
JSON line as the last line of stdout, and fail with a clear message when the box has fewer GPUs than ranks.

With >= 2 visible GPUs the ranks take one GPU each over RCCL; on the 1-GPU test boxes GS_BENCH_SHARE_GPU=1 puts both ranks on
cuda:0 with the exchanges on gloo (control flow only).  The torch.distributed.run form of the contract is covered too.
"""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FAST = ["--steps", "2", "--warmup", "1", "--scene-grid", "1", "--no-cpu-baseline", "--no-extras"]


def _env(share):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.pop("GS_BENCH_SHARE_GPU", None)
    if share:
        env["GS_BENCH_SHARE_GPU"] = "1"
    return env


def _last_json(stdout):
    lines = [ln for ln in stdout.strip().splitlines() if ln.strip()]
    assert lines, "no stdout"
    return json.loads(lines[-1])  # the LAST line must be the JSON record


def _check(rec, n):
    assert rec["n_gpus"] == n and rec["steps"] == 2 and rec["warmup"] == 1
    assert rec["unit"] == "Msplats/s" and rec["value"] > 0 and rec["scaling"] == "weak"
    par = rec["config"]["parallelism"]
    assert ("camera-sharded" in par) or ("gaussian-sharded" in par), par
    assert rec["config"]["dp_mode"] in ("camera", "camera_sparse", "gaussian", "gaussian_dense")
    assert rec["wire"] is not None and rec["wire"]["bytes_out_per_rank_per_step"] > 0
    assert "roofline" in rec and rec["roofline"]["frac"] > 0


def test_driver_command_form_two_ranks():
    share = torch.cuda.device_count() < 2
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + FAST, cwd=ROOT, env=_env(share), capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    _check(_last_json(r.stdout), 2)


def test_torchrun_form_two_ranks():
    share = torch.cuda.device_count() < 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29671", "bench.py", "--gpus", "2"] + FAST
    r = subprocess.run(cmd, cwd=ROOT, env=_env(share), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    _check(_last_json(r.stdout), 2)


def test_refuses_more_ranks_than_gpus():
    n = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, "bench.py", "--gpus", str(n)] + FAST, cwd=ROOT, env=_env(False), capture_output=True,
                       text=True, timeout=300)
    assert r.returncode != 0
    assert "visible GPUs" in r.stderr and not r.stdout.strip()


def test_dead_rank_fails_the_launch():
    """A rank that dies (here: rank 1 told to exit at start-up, in every attempt) must end the launch with a non-zero code, not
    hang rank 0 in its first collective."""
    share = torch.cuda.device_count() < 2
    env = _env(share)
    env["GS_BENCH_TEST_KILL_RANK"] = "1"
    env["GS_BENCH_TEST_KILL_ALWAYS"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + FAST, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode != 0
    assert "rank 1 exited" in r.stderr and "retrying once with --dp-mode camera" in r.stderr


def test_failed_auto_run_falls_back_to_camera_mode():
    """The calibrated run dies (rank 1, first attempt only): ONE retry with --dp-mode camera delivers the line, marked as such."""
    share = torch.cuda.device_count() < 2
    env = _env(share)
    env["GS_BENCH_TEST_KILL_RANK"] = "1"
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + FAST, cwd=ROOT, env=env, capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = _last_json(r.stdout)
    _check(rec, 2)
    assert rec["config"]["dp_mode"] == "camera" and "launch_fallback" in rec["config"]


@pytest.mark.parametrize("form", ["reference", "full"])
def test_dynamic_bench_one_and_two_ranks(form):
    """``bench.py --dynamic`` (BASELINE config 5's frame step): the one-GPU line carries roofline + streaming rows, and the driver's
    ``--gpus 2`` form shards frames round-robin (each rank its own timestamp) and sums the parameter gradients."""
    args = ["--dynamic", "--dynamic-form", form, "--dynamic-splats", "150000", "--steps", "2", "--warmup", "1", "--min-timed-s", "0",
            "--ramp-s", "0"]
    r = subprocess.run([sys.executable, "bench.py"] + args, cwd=ROOT, env=_env(False), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    rec = _last_json(r.stdout)
    assert rec["n_gpus"] == 1 and rec["unit"] == "Msplats/s" and rec["value"] > 0 and "config 5" in rec["config"]["workload"]
    assert rec["roofline"]["frac"] > 0 and rec["roofline"]["whole_step"]["frac"] > 0
    want = "gs_projection_rows_dyn_fwd" if form == "full" else "gs_temporal_slice_fwd"
    assert want in rec["roofline"]["streaming"], sorted(rec["roofline"]["streaming"])
    share = torch.cuda.device_count() < 2
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2"] + args, cwd=ROOT, env=_env(share), capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    rec2 = _last_json(r.stdout)
    assert rec2["n_gpus"] == 2 and rec2["value"] > 0 and "frames round-robin over 2" in rec2["config"]["parallelism"]
