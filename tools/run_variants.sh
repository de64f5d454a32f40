#!/bin/bash
# the bench (FAST and STRICT) over every build of the library found as kajo_amd/libkajo_hip*.so (KAJO_HIP_LIB): the product,
# the experiment libraries of `make -C kajo_amd/csrc experiments` (round-2 kernels and their timing variants, deferred shading)
ls kajo_amd/libkajo_hip_r02.so > /dev/null 2>&1 || echo "(no experiment libraries: make -C kajo_amd/csrc experiments)" >&2
for lib in kajo_amd/libkajo_hip.so kajo_amd/libkajo_hip_r02*.so kajo_amd/libkajo_hip_exp.so; do
  [ -f "$lib" ] || continue
  echo "== $lib"
  for mode in "" "--strict"; do
  KAJO_HIP_LIB=$PWD/$lib python bench.py $mode --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('  %-8s %9.1f Msamples/s  kernel %6.2f ms' % ('$mode' or 'fast', d['value'], d['roofline']['kernel_ms_per_launch']))"
  done
done
