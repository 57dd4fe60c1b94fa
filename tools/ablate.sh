#!/bin/bash
# Build ablation variants of the compositing kernels (debug only): build_abl/libgsplat_hip_abl<N>.so
set -e
cd "$(dirname "$0")/../gscodec_studio_amd/csrc"
mkdir -p ../../build_abl
for n in "$@"; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -fno-slp-vectorize -Wno-unused-function -DGS_ABL=$n $EXTRA -c rasterize.hip -o ../../build_abl/rasterize_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../../build_abl/libgsplat_hip_abl$n.so capi.o projection.o sh.o isect.o radix_sort.o rasterize_ref.o ../../build_abl/rasterize_abl$n.o quantize.o entropy.o unfused.o dynamic.o exchange.o
done
