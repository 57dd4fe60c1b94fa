#!/bin/bash
# Kernel-time profile of ANY command on the GPU box: tools/prof_cmd.sh <tag> <command...> -> gpurun_out/prof_<tag>_kernel_stats.csv
set -u
tag=${1:-x}; shift
root=${GRAFT_REPO_ROOT:-$(pwd)}
out=/tmp/prof_$tag
mkdir -p "$root/gpurun_out" "$out"
cd /tmp && export TMPDIR=/tmp
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out" -o p -- "$@" > "$out/cmd.log" 2>&1 < /dev/null
echo "rocprof rc=$?"; tail -4 "$out/cmd.log"
f=$(find "$out" -name "*kernel_stats.csv" 2>/dev/null | head -1)
[ -n "$f" ] && cp "$f" "$root/gpurun_out/prof_${tag}_kernel_stats.csv" && python -c "
import csv, sys
for r in list(csv.DictReader(open(sys.argv[1])))[:22]:
    print(f\"{r['Name'][:80]:80s} calls {r['Calls']:>6s} avg_us {float(r['AverageNs'])/1e3:9.1f} tot_ms {float(r['TotalDurationNs'])/1e6:8.2f} pct {r['Percentage']}\")
" "$f" < /dev/null
t=$(find "$out" -name "*kernel_trace.csv" 2>/dev/null | head -1)
[ -n "$t" ] && python -c "
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'projection_fwd_kernel' in r['Kernel_Name'] or 'projection_dyn_fwd_kernel' in r['Kernel_Name']]
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
