"""Build profiles/rNN_pmc_traffic.json (driven by tools/pmc.sh) from two rocprofv3 counter-collection CSVs (separate --pmc passes).

    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [<sq_insts_valu.csv> [<mfma_reduce_ab.txt>]]

bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: both counters are in KB, and FETCH_SIZE reports half of the
fetched bytes on gfx950 (MI355X_MICROARCH.md, HBM / rocprofv3 section)."""
import csv
import hashlib
import json
import os
import sys
from collections import defaultdict

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gscodec_studio_amd", "csrc")


def source_hashes():
    """SHA-256 prefix of every kernel source: bench.py reports the counters only while the sources they were taken from are
    unchanged (a kernel edited after the profile makes its traffic figure stale)."""
    out = {}
    for name in sorted(os.listdir(CSRC)):
        if name.endswith((".hip", ".h")):
            out[name] = hashlib.sha256(open(os.path.join(CSRC, name), "rb").read()).hexdigest()[:16]
    return out


def load(path, counter):
    acc = defaultdict(lambda: defaultdict(float))  # kernel -> dispatch -> value (summed over the rows of one dispatch)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    return {k: sum(v.values()) / len(v) for k, v in acc.items()}


def main():
    fetch, write, out = sys.argv[1:4]
    f, w = load(fetch, "FETCH_SIZE"), load(write, "WRITE_SIZE")
    valu = load(sys.argv[4], "SQ_INSTS_VALU") if len(sys.argv) > 4 else {}
    kernels = {}
    for k in f:
        fk, wk = f[k], w.get(k, 0.0)
        kernels[k] = {"FETCH_SIZE_KB_avg": fk, "WRITE_SIZE_KB_avg": wk, "traffic_bytes_per_launch": (2.0 * fk + wk) * 1024.0}
        if k in valu:  # wave-level VALU instructions per launch (third pass: --pmc SQ_INSTS_VALU)
            kernels[k]["valu_wave_instr_per_launch"] = valu[k]
    kernels = dict(sorted(kernels.items(), key=lambda kv: -kv[1]["traffic_bytes_per_launch"]))
    # the VALU issue rate measured in the SAME session (tools/mfma_reduce_ab.hip, first line: 64 independent v_fma_f32 x 2048
    # iterations x 5 waves per SIMD on every SIMD): bench.py's roofline.valu reads it from here
    issue = None
    if len(sys.argv) > 5 and os.path.exists(sys.argv[5]):
        import re

        m = re.search(r"^64 v_fma\s+([0-9.]+) ms", open(sys.argv[5]).read(), re.M)
        if m:
            ms = float(m.group(1))
            issue = {"wave_instr_per_s": 256 * 4 * 5 * 64 * 2048 / (ms * 1e-3), "ms": ms,
                     "source": "tools/mfma_reduce_ab.hip: 64 independent v_fma_f32 x 2048 iterations x 5 waves per SIMD x 1024 SIMDs"}
    json.dump({
        "valu_measured_issue_rate": issue,
        "workload_key": os.environ.get("PMC_WORKLOAD_KEY", "grid3_1920x1080_sh3"),
        "bench_args": os.environ.get("PMC_BENCH_ARGS", ""),
        "command": "tools/pmc.sh: cd /tmp && rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras ; same with --pmc WRITE_SIZE and --pmc SQ_INSTS_VALU (separate passes)",
        "correction": "bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: KB units; FETCH_SIZE reports 1/2 of the fetched bytes on gfx950 (MI355X_MICROARCH.md, HBM section); WRITE_SIZE uncalibrated",
        "source_hashes": source_hashes(),
        "kernels": kernels,
    }, open(out, "w"), indent=1)
    for k, v in list(kernels.items())[:8]:
        print(f"{v['traffic_bytes_per_launch'] / 1e6:9.1f} MB  {k[:90]}")


if __name__ == "__main__":
    main()
