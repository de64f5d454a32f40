#!/bin/bash
# Register / spill / LDS / occupancy figures of every kernel of the two numerics builds, from the compiler's own
# resource-usage remarks (no GPU needed). Usage: tools/kernel_resources.sh [extra hipcc flags, e.g. -DKAJO_PROFILE]
HERE=$(cd "$(dirname "$0")/.." && pwd)
C=$HERE/kajo_amd/csrc
for k in fast strict exact; do
  FP=$([ $k = fast ] && echo -ffp-contract=fast || echo -ffp-contract=off)
  hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$HERE/include -I$C -fno-slp-vectorize $FP \
     "$@" -Rpass-analysis=kernel-resource-usage -c $C/kernel_$k.hip -o /dev/null 2>&1 |
  sed -e 's/ \[-Rpass-analysis=kernel-resource-usage\]//' | awk '/remark: Function Name:/ {name=$NF} /remark: +VGPRs:/ {v=$NF} /SGPRs Spill:/ {ss=$NF} /VGPRs Spill:/ {vs=$NF} /ScratchSize/ {sc=$NF} /Occupancy/ {oc=$NF} /LDS Size/ {printf "%-28s VGPRs %4s  SGPR spills %4s  VGPR spills %4s  scratch %4s B  occupancy %s\n", name, v, ss, vs, sc, oc}'
done
