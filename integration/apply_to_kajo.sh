#!/bin/bash
# Add the "-r hip" backend to a Kajo checkout (skyostil/kajo): one command, nothing of Kajo rewritten.
#
#   integration/apply_to_kajo.sh /path/to/kajo [/path/to/this/repo]
#
# What it does (renderer/Main.cpp:2-4,135-142 and renderer/CMakeLists.txt:9-61,63-72 are the only reference files touched):
#   renderer/hip/Scheduler.{h,cpp}  <- kajo_amd/host/HipScheduler.{h,cpp}, include lines pointed at Kajo's own headers
#   renderer/Main.cpp               += #include "hip/Scheduler.h"  and the  else if (rendererName == "hip")  arm
#   renderer/CMakeLists.txt         += hip/Scheduler.{cpp,h} in the sources, this repo's include/ and libkajo_hip + rccl + amdhip64
# Idempotent: a second run changes nothing. Build this repo first (python -c "import __graft_entry__ as g; g.build()").
set -euo pipefail
KAJO=${1:?usage: apply_to_kajo.sh /path/to/kajo [/path/to/kajo-hip]}
HERE=${2:-$(cd "$(dirname "$0")/.." && pwd)}
R="$KAJO/renderer"
[ -f "$R/Main.cpp" ] && [ -f "$R/CMakeLists.txt" ] || { echo "not a Kajo checkout: $KAJO" >&2; exit 1; }

mkdir -p "$R/hip"
# the backend itself: same text, Kajo's include style (cpu/Scheduler.h includes "renderer/Scheduler.h")
sed -e 's|#include "Scheduler.h"|#include "renderer/Scheduler.h"|' -e 's|KAJO_HIP_SCHEDULER_H|HIP_SCHEDULER_H|g' \
    "$HERE/kajo_amd/host/HipScheduler.h" > "$R/hip/Scheduler.h"
sed -e 's|#include "HipScheduler.h"|#include "Scheduler.h"|' -e 's|#include "Image.h"|#include "renderer/Image.h"|' \
    -e 's|#include "Preview.h"|#include "renderer/Preview.h"|' "$HERE/kajo_amd/host/HipScheduler.cpp" > "$R/hip/Scheduler.cpp"

if ! grep -q '"hip/Scheduler.h"' "$R/Main.cpp"; then
    sed -i -e 's|^#include "gl/Scheduler.h"$|&\n#include "hip/Scheduler.h"|' \
           -e 's|^\(\s*\)scheduler.reset(new gl::Scheduler(scene, image.get(), preview.get()));$|&\n\1} else if (rendererName == "hip") {\n\1scheduler.reset(new hip::Scheduler(scene, image.get(), preview.get()));|' "$R/Main.cpp"
fi
grep -q '"hip/Scheduler.h"' "$R/Main.cpp" && grep -q 'new hip::Scheduler' "$R/Main.cpp" || { echo "Main.cpp: anchors not found" >&2; exit 1; }

if ! grep -q 'hip/Scheduler.cpp' "$R/CMakeLists.txt"; then
    sed -i -e 's|^\(\s*\)gl/SurfaceShader.h$|&\n\n\1# HIP renderer (MI355X): kajo-hip\n\1hip/Scheduler.cpp\n\1hip/Scheduler.h|' \
           -e 's|^\(\s*\)${CMAKE_THREAD_LIBS_INIT}$|&\n\1kajo_hip\n\1rccl\n\1amdhip64|' "$R/CMakeLists.txt"
    cat >> "$R/CMakeLists.txt" <<CM

# kajo-hip: the C ABI (include/kajo_hip.h, libkajo_hip.so) and the ROCm runtime it links
target_include_directories(renderer PRIVATE "$HERE/include" /opt/rocm/include)
target_link_directories(renderer PRIVATE "$HERE/kajo_amd" /opt/rocm/lib)
target_compile_definitions(renderer PRIVATE __HIP_PLATFORM_AMD__)
CM
fi
grep -q 'hip/Scheduler.cpp' "$R/CMakeLists.txt" && grep -q 'kajo_hip' "$R/CMakeLists.txt" || { echo "CMakeLists.txt: anchors not found" >&2; exit 1; }
echo "kajo-hip backend added to $KAJO: renderer -r hip scene.json"
