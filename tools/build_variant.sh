#!/bin/bash
# Build a VARIANT of the library (tools only; never what bench.py times): kajo_amd/variants/libkajo_hip_<name>.so, its capi / stage
# compiled -DKAJO_TUNING like libkajo_hip_tune.so, the kernel translation units with the given make variables.
# usage: tools/build_variant.sh <name> [MAKEVAR=value ...]     e.g.  tools/build_variant.sh w5 KFLAGS=-DKAJO_WAVES_PER_SIMD=5
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p kajo_amd/variants
make -s -j8 -C kajo_amd/csrc OUT=$PWD/kajo_amd/variants/libkajo_hip_$name.so BUILD=$PWD/kajo_amd/csrc/build_var_$name HOSTFLAGS=-DKAJO_TUNING "$@" && echo "built variant $name"
