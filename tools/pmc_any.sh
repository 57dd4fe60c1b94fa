#!/bin/bash
# Arbitrary SQ/LDS counters per kernel on the GPU box, one rocprofv3 --pmc pass per counter (no trace domains besides
# --kernel-trace):  tools/pmc_any.sh <tag> COUNTER [COUNTER...]   ->  gpurun_out/pmc_<tag>.txt (per-kernel averages)
set -u
tag=$1; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
mkdir -p "$root/gpurun_out"
cd /tmp && export TMPDIR=/tmp
for c in "$@"; do
    rm -rf /tmp/pmca_$c
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmca_$c -o p -- \
        ${PMC_CMD:-python "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-dp-projection --min-timed-s 0} > /tmp/pmca_$c.log 2>&1 < /dev/null
    echo "$c rc=$?"
done
cat > /tmp/pmca_sum.py <<'PY' 
import csv, glob, sys
from collections import defaultdict
out, counters = sys.argv[1], sys.argv[2:]
res = defaultdict(dict)
for c in counters:
    fs = glob.glob(f"/tmp/pmca_{c}/**/*counter_collection.csv", recursive=True)
    if not fs:
        print("no csv for", c); continue
    acc = defaultdict(lambda: defaultdict(float))
    for r in csv.DictReader(open(fs[0])):
        if r["Counter_Name"] == c:
            acc[r["Kernel_Name"]][r["Dispatch_Id"]] += float(r["Counter_Value"])
    for k, v in acc.items():
        res[k][c] = sum(v.values()) / len(v)
with open(out, "w") as f:
    for k, v in sorted(res.items(), key=lambda kv: -max(kv[1].values())):
        line = f"{k[:80]:80s} " + " ".join(f"{c}={v.get(c, float('nan')):.4g}" for c in counters)
        print(line); f.write(line + "\n")
PY
python /tmp/pmca_sum.py "$root/gpurun_out/pmc_$tag.txt" "$@" < /dev/null
