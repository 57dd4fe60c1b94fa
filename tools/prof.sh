#!/bin/bash
# Kernel-time profile of bench.py on the GPU box:  tools/prof.sh <tag> [bench args...]
# Writes gpurun_out/prof_<tag>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, CSV output) and prints its head.
# Everything is bounded by `timeout`; no command reads stdin.
set -u
tag=${1:-x}; shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=/tmp/prof_$tag
mkdir -p "$root/gpurun_out" "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- \
    python "$root/bench.py" --steps 20 --warmup 5 --no-cpu-baseline --no-extras --no-dp-projection --min-timed-s 0 "$@" > "$out/bench.log" 2>&1 < /dev/null
echo "rocprof rc=$?"
f=$(find "$out" -name "*kernel_stats.csv" 2>/dev/null | head -1)
if [ -n "$f" ]; then
    cp "$f" "$root/gpurun_out/prof_${tag}_kernel_stats.csv"
    python -c "
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:24]:
    print(f\"{r['Name'][:90]:90s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} min {float(r['MinNs'])/1e3:8.1f} pct {r['Percentage']}\")
" "$f" < /dev/null
else
    echo "no kernel_stats.csv under $out"; tail -5 "$out/bench.log"
fi
# timeline of the LAST step (kernel name, start offset and duration in us) -> gpurun_out/prof_<tag>_last_step.txt
t=$(find "$out" -name "*kernel_trace.csv" 2>/dev/null | head -1)
if [ -n "$t" ]; then
    python -c "
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# one step = from one projection_fwd_kernel to the next
idx = [i for i, n in enumerate(names) if 'projection_fwd_kernel' in n or 'projection_dyn_fwd_kernel' in n]
a, b = idx[-2], idx[-1]
t0 = int(rows[a]['Start_Timestamp'])
with open(sys.argv[2], 'w') as f:
    prev_end = t0
    for r in rows[a:b]:
        s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        f.write(f\"{(s - t0) / 1e3:9.1f} us  +gap {(s - prev_end) / 1e3:6.1f}  dur {(e - s) / 1e3:8.1f}  {r['Kernel_Name'][:110]}\\n\")
        prev_end = e
    f.write(f\"step length {(int(rows[b]['Start_Timestamp']) - t0) / 1e3:.1f} us, {b - a} launches\\n\")
" "$t" "$root/gpurun_out/prof_${tag}_last_step.txt" < /dev/null
    tail -1 "$root/gpurun_out/prof_${tag}_last_step.txt"
fi
