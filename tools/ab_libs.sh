#!/bin/bash
# A/B of library builds on the GPU box: tools/ab_libs.sh [rounds] -- runs bench.py with the in-tree library and with every
# build_ab/*.so (GSPLAT_HIP_LIB), interleaved round-robin, and prints ms/step per run.  Variant libraries are built by hand
# with a -D macro (build_ab/ is git-ignored through *.so and travels with the gpurun snapshot).
rounds=${1:-2}
root=${GRAFT_REPO_ROOT:-$(pwd)}
shift || true
for r in $(seq 1 "$rounds"); do
    for lib in default "$root"/build_ab/*.so; do
        if [ "$lib" = default ]; then unset GSPLAT_HIP_LIB; else export GSPLAT_HIP_LIB=$lib; fi
        ms=$(timeout 300 python "$root/bench.py" --no-cpu-baseline --no-extras --no-dp-projection "$@" 2>/dev/null < /dev/null |
             python -c "import sys,json; print('%.4f' % json.loads(sys.stdin.read())['ms_per_step'])")
        echo "$(basename "$lib") $ms"
    done
done
