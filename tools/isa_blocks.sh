#!/bin/bash
# Compile-only: the VALU mix (issue classes of tools/valu_rate.hip) of each block of a render kernel's loop, from the compiler's
# assembly with the block boundaries of tools/blockprof.py written into it as comments (-DKAJO_MARKS; not a build that runs).
# Blocks are taken in LAYOUT order between consecutive marks; a loop inside a block is counted once (static counts).
# usage: tools/isa_blocks.sh [fast|strict] [kernel name, default kajo_render_<mode>]
HERE=$(cd "$(dirname "$0")/.." && pwd); C=$HERE/kajo_amd/csrc; K=${1:-fast}; KERNEL=${2:-kajo_render_$K}
FP=$([ $K = fast ] && echo "-ffp-contract=fast" || echo -ffp-contract=off)
S=$(mktemp /tmp/isa_blocks.XXXXXX.s)
hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -I$HERE/include -I$C -fno-slp-vectorize $FP -DKAJO_MARKS -S --cuda-device-only -o $S $C/kernel_$K.hip 2>/dev/null || exit 1
NAMES=("camera-ray block (after mark 4)" "traversal" "vertex / shadow-result" "light + BSDF" "tail" "light loop: samples + own part" "light loop: helpers, rays fetched" "light loop: lists walked" "light loop: contributions")
awk -v k="$KERNEL:" '$1 == k {f = 1} f {n++; if ($0 ~ /; KMARK/) print n - 1, $NF; if ($0 ~ /s_endpgm/) {print n - 1, "end"; exit}}' $S > $S.marks
prev=""; prevk=""
while read line mark; do
  if [ -n "$prev" ]; then
    case $prevk in 4) name=${NAMES[0]};; 0) name=${NAMES[1]};; 1) name=${NAMES[2]};; 2) name=${NAMES[3]};; 3) name=${NAMES[4]};; 5) name=${NAMES[6]};; 6) name=${NAMES[7]};; 7) name=${NAMES[8]};; 8) name=${NAMES[5]};; esac
    echo "== after mark $prevk up to mark $mark: $name"
    python3 $HERE/tools/isa_classes.py $S $KERNEL $prev $line
  fi
  prev=$line; prevk=$mark
done < $S.marks
rm -f $S $S.marks
