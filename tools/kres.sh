#!/bin/bash
# usage: tools/kres.sh <file.hip> [name-filter] -- VGPRs / spills / occupancy of every kernel in one csrc file
cd /root/repo/3pu_pytorch_amd/csrc
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -munsafe-fp-atomics -c "$1" -o /tmp/kres.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | grep -E "error|Name:|VGPRs:|AGPRs:|VGPRs Spill|Occupancy|ScratchSize|LDS Size" \
  | sed 's/\[-Rpass.*//; s/^[^ ]* //; s/remark: *//' | paste - - - - - - - | grep -E "error|${2:-.}" | cut -c1-330
