#!/bin/bash
# Re-tune the launch knobs of the compositing kernels on the bench workload: tools/sweep_tuning.sh > gpurun_out/sweep.txt
# One bench.py run per setting (the values travel through GS_RASTER_* -> gs_raster_plan).
root=${GRAFT_REPO_ROOT:-$(pwd)}
run() {
    ms=$(env "$@" python "$root/bench.py" --no-cpu-baseline --no-extras --steps 30 --warmup 10 --min-timed-s 0.3 2>/dev/null | tail -1 |
         python -c "import sys, json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
    echo "$ms ms/step  $*"
}
run GS_NONE=1
for v in 128 192 320 384; do run GS_RASTER_SEG=$v; done
for v in 1024 1536 3072 4096; do run GS_RASTER_SOLO=$v; done
for v in 4 8 32 64; do run GS_RASTER_XCD_FWD=$v; done
for v in 4 8 32 64; do run GS_RASTER_XCD_BWD=$v; done
