"""Host time per phase of one gaussian-sharded (world 1, collectives forced through RCCL) or plain step at a launch-bound size:
perf_counter accumulators around the Python entry points (forward) and the autograd nodes' backward methods."""
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

DIST = os.environ.get("DIST", "1") == "1"
dev = torch.device("cuda:0")
if DIST:
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29734")
    os.environ["GS_DIST_FORCE_COLLECTIVES"] = "1"
    torch.cuda.set_device(dev)
    torch.distributed.init_process_group("nccl", rank=0, world_size=1)
from gscodec_studio_amd import _step, _wrapper, distributed, rasterization, rendering  # noqa: E402
from gscodec_studio_amd._helper import sh_workload  # noqa: E402

acc = {}


def timed(obj, name, label=None):
    fn = getattr(obj, name)
    label = label or name

    def wrap(*a, **k):
        t0 = time.perf_counter()
        try:
            return fn(*a, **k)
        finally:
            acc[label] = acc.get(label, 0.0) + time.perf_counter() - t0
    setattr(obj, name, wrap)


timed(distributed, "gather_shard_meta")
timed(rendering, "project_rows")
timed(distributed, "exchange_rows")
timed(distributed, "_all_to_all_single")
timed(_step, "rows_begin")
timed(_step, "rows_composite")
timed(distributed, "exchange_overflowed")
for cls, lab in ((_wrapper._ProjectRows, "bwd project_rows"), (distributed._ExchangeRows, "bwd exchange"),
                 (_step._StepRowsComposite, "bwd composite"), (_step._StepComposite, "bwd composite1"), (_step._StepProject, "bwd project1")):
    b = cls.backward

    def mk(b=b, lab=lab):
        def wrap(*a, **k):
            t0 = time.perf_counter()
            try:
                return b(*a, **k)
            finally:
                acc[lab] = acc.get(lab, 0.0) + time.perf_counter() - t0
        return staticmethod(wrap)
    cls.backward = mk()

w = sh_workload(scene_grid=1, device=dev)
n = int(os.environ.get("N", "3000"))
params = {k: w[k][:n].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh")}
vm, Ks = w["viewmats"][:1].contiguous(), w["Ks"][:1].contiguous()
tf = tb = 0.0


def step():
    global tf, tb
    for p in params.values():
        p.grad = None
    t0 = time.perf_counter()
    rc, ra, meta = rasterization(params["means"], params["quats"], params["scales"], params["opacities"], params["sh"], vm, Ks,
                                 1920, 1080, sh_degree=3, packed=False, distributed=DIST)
    t1 = time.perf_counter()
    rc.sum().backward()
    t2 = time.perf_counter()
    tf += t1 - t0
    tb += t2 - t1


for _ in range(30):
    step()
torch.cuda.synchronize()
acc.clear()
tf = tb = 0.0
K = 300
t0 = time.perf_counter()
for _ in range(K):
    step()
torch.cuda.synchronize()
tot = time.perf_counter() - t0
print(f"step {1e6 * tot / K:.0f} us: forward {1e6 * tf / K:.0f}, loss + backward {1e6 * tb / K:.0f}")
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print(f"  {k:28s} {1e6 * v / K:7.1f} us")
if os.environ.get("PROFILE") == "1":  # Python-level profile of the same steps (which calls the host time goes to)
    import cProfile
    import io
    import pstats

    pr = cProfile.Profile()
    pr.enable()
    for _ in range(100):
        step()
    pr.disable()
    torch.cuda.synchronize()
    s = io.StringIO()
    pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(32)
    print(s.getvalue()[:6500])
