#!/bin/bash
# HBM traffic + VALU instruction counters of bench.py's kernels on the GPU box:  tools/pmc.sh <out.json>
# (PMC_BENCH_ARGS="--dynamic ..." PMC_WORKLOAD_KEY=... for another workload of bench.py)
# Three SEPARATE rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE, SQ_INSTS_VALU), as MI355X_MICROARCH.md prescribes; the JSON
# (per-launch bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 and wave-level VALU instructions) is built by tools/pmc_traffic.py.
set -u
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=${1:-$root/gpurun_out/pmc_traffic.json}
case "$out" in /*) ;; *) out="$root/$out" ;; esac
mkdir -p "$(dirname "$out")"
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
    rm -rf /tmp/pmc_$c
    timeout -k 5 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- \
        python "$root/bench.py" --steps 3 --warmup 1 --no-cpu-baseline --no-extras --no-dp-projection --min-timed-s 0 --ramp-s 0 ${PMC_BENCH_ARGS:-} > /tmp/pmc_$c.log 2>&1 < /dev/null
    echo "$c rc=$?"
done
f=$(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
w=$(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
v=$(find /tmp/pmc_SQ_INSTS_VALU -name "*counter_collection.csv" | head -1)
if [ -n "$f" ] && [ -n "$w" ] && [ -n "$v" ]; then
    # (the measured VALU issue rate of this session rides along: bench.py's roofline.valu reads it from the JSON)
    [ -x "$root/build_abl/mfma_reduce_ab" ] && timeout 120 "$root/build_abl/mfma_reduce_ab" > /tmp/mfma_reduce_ab.txt 2>&1
    python "$root/tools/pmc_traffic.py" "$f" "$w" "$out" "$v" /tmp/mfma_reduce_ab.txt < /dev/null
else
    echo "missing counter CSVs: '$f' '$w' '$v'"; tail -3 /tmp/pmc_FETCH_SIZE.log
fi
