#!/bin/bash
# A/B on the GPU box: rebuild the kernels with each KFLAGS variant and run the bench (no CPU legs).
# usage: tools/ab.sh "name1|flags1" "name2|flags2" ...
cd "$(dirname "$0")/.."
for v in "$@"; do
  name="${v%%|*}"; flags="${v#*|}"
  touch kajo_amd/csrc/kernel_fast.hip
  make -s -C kajo_amd/csrc KFLAGS="$flags" >/dev/null 2>&1 || { echo "$name BUILD FAILED"; continue; }
  for rep in 1 2; do
  python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().split('\n')[-1])
print('%-28s %9.1f Msamples/s  %6.2f ms/step  kernel %6.2f ms  frac %.4f' % ('$name', d['value'], d['ms_per_step'], d['roofline']['kernel_ms_per_launch'], d['roofline']['frac']))"
  done
done
touch kajo_amd/csrc/kernel_fast.hip
make -s -C kajo_amd/csrc >/dev/null 2>&1
