#!/bin/bash
# Compile-only: registers and spills of the main render kernels at a given waves-per-SIMD bound.
# usage: tools/vgpr_check.sh [waves per SIMD, default 5] [fast|strict]
HERE=$(cd "$(dirname "$0")/.." && pwd); C=$HERE/kajo_amd/csrc; W=${1:-5}; K=${2:-fast}
FP=$([ $K = fast ] && echo -ffp-contract=fast || echo -ffp-contract=off)
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$HERE/include -I$C -fno-slp-vectorize $FP \
  -DKAJO_WAVES_PER_SIMD=$W -Rpass-analysis=kernel-resource-usage -c $C/kernel_$K.hip -o /dev/null 2>&1 | sed -e 's/ \[-Rpass-analysis=kernel-resource-usage\]//' |
  awk '/remark: Function Name:/ {name=$NF} /remark: +VGPRs:/ {v=$NF} /SGPRs Spill:/ {ss=$NF} /VGPRs Spill:/ {vs=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {oc=$NF} /LDS Size/ {if (name ~ /render/) printf "%-28s VGPRs %4s  SGPR spills %4s  VGPR spills %4s  scratch %4s B  occupancy %s\n", name, v, ss, vs, sc, oc}'
