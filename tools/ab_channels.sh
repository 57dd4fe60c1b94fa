#!/bin/bash
# A/B of library builds on the wide-channel step: tools/ab_channels.sh [rounds] [D ...] -- tools/bench_channels.py with the
# in-tree library and with every build_ab/*.so (GSPLAT_HIP_LIB), interleaved round-robin.
rounds=${1:-2}
shift || true
root=${GRAFT_REPO_ROOT:-$(pwd)}
for r in $(seq 1 "$rounds"); do
    for lib in default "$root"/build_ab/*.so; do
        if [ "$lib" = default ]; then unset GSPLAT_HIP_LIB; else export GSPLAT_HIP_LIB=$lib; fi
        timeout 300 python "$root/tools/bench_channels.py" "${@:-9}" 2>/dev/null < /dev/null | grep "^D" | sed "s|^|$(basename "$lib") |"
    done
done
