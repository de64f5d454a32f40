#!/bin/bash
# CPU only: kajo_amd/csrc/stage.cpp (object records, uniform grid, per-light visibility lists) under AddressSanitizer + UBSan on the scenes
# of the tests (spheres.json, the dialect scene, 120 / 300 / 1000-sphere scenes, the adversarial geometry of tests/test_shadow_lists_cpu.py).
# The same harness (tools/host_san.cpp) runs as a test with the loader, the launch order and the tile map: tests/test_sanitize_cpu.py.
HERE=$(cd "$(dirname "$0")/.." && pwd) || exit 1
cd "$HERE" && exec python3 -m pytest -x -q tests/test_sanitize_cpu.py "$@"
