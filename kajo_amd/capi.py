"""ctypes binding of libkajo_hip.so (include/kajo_hip.h). Host plumbing only.

The library is the product's compute path; there is no Python or CPU fallback: a missing or
unloadable library raises ImportError, a missing GPU makes kajo_hip_create fail.
"""
from __future__ import annotations

import ctypes as C
import os

# PyTorch ships its own copy of the HIP runtime (torch/lib/libamdhip64.so, soname
# libamdhip64.so.7). Two HIP runtimes in one process do not share the GPU, so torch must be
# imported BEFORE libkajo_hip.so: the dynamic loader then satisfies the library's
# libamdhip64.so.7 dependency with the copy torch already mapped, and both use one runtime.
import torch  # noqa: F401  (plumbing: device memory, streams, torch.distributed)

from .scene import KajoScene

_HERE = os.path.dirname(os.path.abspath(__file__))
# KAJO_HIP_LIB: a diagnostic twin of the library (e.g. the -DKAJO_PROFILE build of `make prof`); never a CPU path
LIB_PATH = os.environ.get("KAJO_HIP_LIB") or os.path.join(_HERE, "libkajo_hip.so")

KAJO_OK, KAJO_E_INVALID, KAJO_E_HIP, KAJO_E_NO_DEVICE, KAJO_E_STATE = 0, -1, -2, -3, -4

KAJO_FLAG_FAST = 0  # (neither of the two below)
KAJO_FLAG_STRICT = 1
KAJO_FLAG_COUNTERS = 2
KAJO_FLAG_NO_GRID = 4
KAJO_FLAG_NO_REORDER = 8
KAJO_FLAG_NO_SPLIT = 16
KAJO_FLAG_COOP = 32      # experiment library libkajo_hip_r02.so only
KAJO_FLAG_DEFERRED = 64  # experiment library libkajo_hip_exp.so only
KAJO_FLAG_NO_SHADOW_LISTS = 128
KAJO_FLAG_NO_ONE_LIGHT = 256  # every numerics build: the any-number-of-lights instance for a one-light scene (A/B, tests)
KAJO_FLAG_EXACT = 512  # decision-exact numerics: STRICT's decisions, FAST's radiance arithmetic

# every symbol include/kajo_hip.h declares
EXPORTS = [
    "kajo_hip_default_params", "kajo_hip_create", "kajo_hip_destroy", "kajo_hip_render", "kajo_hip_wait",
    "kajo_hip_reset", "kajo_hip_set_pass_count", "kajo_hip_resolve_argb8", "kajo_hip_read_radiance", "kajo_hip_resolve_argb8_device",
    "kajo_hip_tile_buffer", "kajo_hip_compose", "kajo_hip_set_stream", "kajo_hip_counters",
    "kajo_hip_stage_scene", "kajo_hip_last_error", "kajo_hip_version", "kajo_hip_kat_trace", "kajo_hip_kat_shade",
    "kajo_hip_kat_strictmath", "kajo_hip_stage_shadow_lists", "kajo_hip_resolve_gathered_argb8_device", "kajo_hip_stage_info",
    "kajo_hip_launch_order",
]


class KajoStageInfo(C.Structure):
    _fields_ = [("closedRoom", C.c_int32), ("grid", C.c_int32), ("shadowLists", C.c_int32), ("reserved", C.c_int32),
                ("room", C.c_float * 6), ("gridCenter", C.c_float * 3), ("gridReach", C.c_float)]


class KajoParams(C.Structure):
    _fields_ = [
        ("samplesPerPass", C.c_int32),
        ("depthLimit", C.c_int32),
        ("seed", C.c_uint64),
        ("flags", C.c_uint32),
        ("device", C.c_int32),
        ("tileW", C.c_int32),
        ("tileH", C.c_int32),
        ("tileIndex", C.c_int32),
        ("tileCount", C.c_int32),
        ("passesPerLaunch", C.c_int32),
    ]


class KajoCounters(C.Structure):
    _fields_ = [
        ("passes", C.c_uint64),
        ("paths", C.c_uint64),
        ("traversals", C.c_uint64),
        ("vertices", C.c_uint64),
        ("primitiveTests", C.c_uint64),
        ("laneSlots", C.c_uint64),
        ("kernelMs", C.c_double),
        ("launches", C.c_uint64),
        ("shadowQueries", C.c_uint64),  # (appended in round 4; the experiment libraries of earlier rounds leave it 0)
        ("tailGroups", C.c_uint64),  # (appended in round 5)
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C kajo_amd/csrc)" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        L.kajo_hip_last_error.restype = C.c_char_p
        L.kajo_hip_version.restype = C.c_char_p
        L.kajo_hip_create.argtypes = [C.POINTER(KajoScene), C.c_int, C.c_int, C.POINTER(KajoParams), C.POINTER(C.c_void_p)]
        L.kajo_hip_destroy.argtypes = [C.c_void_p]
        L.kajo_hip_render.argtypes = [C.c_void_p, C.c_int]
        L.kajo_hip_wait.argtypes = [C.c_void_p]
        L.kajo_hip_reset.argtypes = [C.c_void_p]
        L.kajo_hip_set_pass_count.argtypes = [C.c_void_p, C.c_int]
        L.kajo_hip_resolve_argb8.argtypes = [C.c_void_p, C.c_void_p]
        L.kajo_hip_read_radiance.argtypes = [C.c_void_p, C.c_void_p]
        L.kajo_hip_resolve_argb8_device.argtypes = [C.c_void_p, C.c_void_p]
        L.kajo_hip_tile_buffer.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.kajo_hip_compose.argtypes = [C.c_void_p, C.c_void_p]
        L.kajo_hip_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.kajo_hip_counters.argtypes = [C.c_void_p, C.POINTER(KajoCounters)]
        L.kajo_hip_stage_scene.argtypes = [C.POINTER(KajoScene), C.c_void_p, C.c_void_p]
        L.kajo_hip_default_params.argtypes = [C.POINTER(KajoParams)]
        if hasattr(L, "kajo_hip_resolve_gathered_argb8_device"):
            L.kajo_hip_resolve_gathered_argb8_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        if hasattr(L, "kajo_hip_stage_shadow_lists"):  # (round 4; the experiment libraries of earlier rounds do not have it)
            L.kajo_hip_stage_shadow_lists.argtypes = [C.POINTER(KajoScene), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_void_p, C.c_void_p,
                                                      C.c_size_t, C.c_void_p, C.c_void_p, C.c_size_t]
        if hasattr(L, "kajo_hip_stage_info"):
            L.kajo_hip_stage_info.argtypes = [C.POINTER(KajoScene), C.POINTER(KajoStageInfo)]
        if hasattr(L, "kajo_hip_launch_order"):  # (round 6)
            L.kajo_hip_launch_order.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_size_t, C.POINTER(C.c_uint32)]
        L.kajo_hip_kat_trace.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 8
        L.kajo_hip_kat_shade.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.kajo_hip_kat_strictmath.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        _lib = L
    return _lib


class KajoError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("kajo_hip error %d: %s" % (code, message))
        self.code = code


def check(rc):
    if rc != 0:
        raise KajoError(rc, (lib().kajo_hip_last_error() or b"").decode())
