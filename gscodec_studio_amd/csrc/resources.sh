#!/bin/bash
# usage: resources.sh file.hip  -- prints kernel name, VGPRs, scratch bytes, occupancy
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -Rpass-analysis=kernel-resource-usage -c "$1" -o /tmp/_res.o 2>&1 \
 | grep -E "Function Name:| VGPRs:|ScratchSize|Occupancy|LDS Size" \
 | sed -E 's/^.*remark: [^ ]+ +//; s/ \[-Rpass.*//' \
 | awk '/Function Name/{if(n)print n, v, s, o, l; n=$3} / VGPRs:|^VGPRs:/{v="vgpr="$2} /ScratchSize/{s="scratch="$4} /Occupancy/{o="occ="$4} /LDS Size/{l="lds="$5} END{print n, v, s, o, l}' \
 | sed 's/_ZN12_GLOBAL__N_1//' | cut -c1-160
