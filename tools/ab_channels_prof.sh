#!/bin/bash
# Kernel-time A/B of library builds on the wide-channel step (rocprofv3 averages of the two compositing kernels):
#   tools/ab_channels_prof.sh D [D ...]
root=${GRAFT_REPO_ROOT:-$(pwd)}
for D in "$@"; do
  for lib in default "$root"/build_ab/*.so; do
    if [ "$lib" = default ]; then unset GSPLAT_HIP_LIB; else export GSPLAT_HIP_LIB=$lib; fi
    tag=ab_$(basename "$lib" .so)_$D
    bash "$root/tools/prof_cmd.sh" $tag python "$root/tools/bench_channels.py" $D > /tmp/$tag.log 2>&1
    echo "== $(basename "$lib") D=$D"; grep -E "raster_(tile_fwd|seg_bwd)" /tmp/$tag.log | cut -c1-60,100-
  done
done
