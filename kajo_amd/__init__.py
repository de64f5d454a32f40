"""kajo_amd -- MI355X-native rendering backend for Kajo's per-pixel Monte-Carlo integrator.

csrc/      HIP kernels + the C ABI (libkajo_hip.so, include/kajo_hip.h)
host/      C++ host side: hip::Scheduler behind Kajo's Scheduler plugin interface, headless driver
scene.py   flat scene model (ctypes mirror of include/kajo_scene.h) + synthetic scenes
renderer.py / capi.py   Python plumbing over the C ABI (tests, bench, multi-GPU gather)
"""
from .scene import Scene  # noqa: F401
