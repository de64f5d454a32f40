"""Host-side driver of the HIP backend: the Python counterpart of hip::Scheduler
(kajo_amd/host/HipScheduler.cpp), which in turn stands where cpu::Scheduler stands in the
reference (renderer/cpu/Scheduler.cpp:53-85: own a renderer, run passes, hand pixels to the
Image). Used by the tests, bench.py and the multi-GPU path; all rendering happens in
libkajo_hip.so.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .scene import Scene


class HipRenderer:
    def __init__(self, scene: Scene, width: int, height: int, spp: int = 32, depth_limit: int = 8,
                 seed: int = 0o715517, strict: bool = False, exact: bool = False, counters: bool = False, device: int = 0,
                 tile=(64, 16), tile_index: int = 0, tile_count: int = 1, passes_per_launch: int = 0, flags: int = 0):
        L = capi.lib()
        self._L = L
        self.scene = scene
        self.width, self.height = int(width), int(height)
        self.n = int(np.sqrt(float(spp)))  # Renderer.cpp:38
        p = capi.KajoParams()
        L.kajo_hip_default_params(C.byref(p))
        p.samplesPerPass = spp
        p.depthLimit = depth_limit
        p.seed = seed
        p.flags = (capi.KAJO_FLAG_STRICT if strict else 0) | (capi.KAJO_FLAG_EXACT if exact else 0) | (capi.KAJO_FLAG_COUNTERS if counters else 0) | int(flags)
        p.device = device
        p.tileW, p.tileH = tile
        p.tileIndex, p.tileCount = tile_index, tile_count
        p.passesPerLaunch = passes_per_launch
        self.params = p
        self._pod = scene.pod()
        h = C.c_void_p()
        capi.check(L.kajo_hip_create(C.byref(self._pod), self.width, self.height, C.byref(p), C.byref(h)))
        self._h = h
        self.passes = 0

    # -- lifecycle -------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            self._L.kajo_hip_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- rendering -------------------------------------------------------------------------
    def render(self, passes: int = 1, wait: bool = False):
        capi.check(self._L.kajo_hip_render(self._h, int(passes)))
        self.passes += int(passes)
        if wait:
            self.wait()
        return self

    def wait(self):
        capi.check(self._L.kajo_hip_wait(self._h))

    def reset(self):
        capi.check(self._L.kajo_hip_reset(self._h))
        self.passes = 0

    def set_pass_count(self, passes_done: int):
        capi.check(self._L.kajo_hip_set_pass_count(self._h, int(passes_done)))
        self.passes = int(passes_done)

    def set_stream(self, stream_ptr: int):
        capi.check(self._L.kajo_hip_set_stream(self._h, C.c_void_p(stream_ptr)))

    # -- outputs ---------------------------------------------------------------------------
    def radiance(self) -> np.ndarray:
        """(H, W, 4) float32: sum over passes of radiance / S (divide by .passes for the estimate)."""
        out = np.empty((self.height, self.width, 4), np.float32)
        capi.check(self._L.kajo_hip_read_radiance(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def argb8(self) -> np.ndarray:
        """(H, W) uint32 ARGB8, sRGB-encoded, row 0 = top (Image::pixels, renderer/Image.h:18-20)."""
        out = np.empty((self.height, self.width), np.uint32)
        capi.check(self._L.kajo_hip_resolve_argb8(self._h, out.ctypes.data_as(C.c_void_p)))
        return out

    def counters(self) -> dict:
        c = capi.KajoCounters()
        capi.check(self._L.kajo_hip_counters(self._h, C.byref(c)))
        return {k: getattr(c, k) for k, _ in capi.KajoCounters._fields_}

    # -- known-answer hooks ----------------------------------------------------------------
    def kat_trace(self, origins, dirs):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        n = o.shape[0]
        idx = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        pos, nor, tan, bi = (np.zeros((n, 3), np.float32) for _ in range(4))
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        capi.check(self._L.kajo_hip_kat_trace(self._h, n, p(o), p(d), p(idx), p(t), p(pos), p(nor), p(tan), p(bi)))
        return dict(idx=idx, t=t, position=pos, normal=nor, tangent=tan, binormal=bi)

    def kat_shade(self, origins, dirs, states):
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        st = np.ascontiguousarray(states, np.uint64)
        n = o.shape[0]
        rgb = np.zeros((n, 3), np.float32)
        fin = np.zeros((n, 2), np.uint64)
        p = lambda a: a.ctypes.data_as(C.c_void_p)
        capi.check(self._L.kajo_hip_kat_shade(self._h, n, p(o), p(d), p(st), p(rgb), p(fin)))
        return rgb, fin

    # -- multi-GPU plumbing ----------------------------------------------------------------
    def tile_buffer(self):
        ptr, nbytes = C.c_void_p(), C.c_size_t()
        capi.check(self._L.kajo_hip_tile_buffer(self._h, C.byref(ptr), C.byref(nbytes)))
        return ptr.value, nbytes.value

    def compose(self, gathered_device_ptr: int):
        capi.check(self._L.kajo_hip_compose(self._h, C.c_void_p(gathered_device_ptr)))


def stage_scene(scene: Scene):
    """Host-only: (inverse+determinant per object [n,17], camera basis [4,3]) as create() stages them."""
    L = capi.lib()
    n = scene.n_planes + scene.n_spheres
    inv = np.zeros((n, 17), np.float32)
    basis = np.zeros((4, 3), np.float32)
    pod = scene.pod()
    capi.check(L.kajo_hip_stage_scene(C.byref(pod), inv.ctypes.data_as(C.c_void_p), basis.ctypes.data_as(C.c_void_p)))
    return inv, basis


def stage_info(scene: Scene) -> dict:
    """Host-only: what create() decides about the scene's culling structures (include/kajo_hip.h KajoStageInfo)."""
    L = capi.lib()
    pod = scene.pod()
    info = capi.KajoStageInfo()
    capi.check(L.kajo_hip_stage_info(C.byref(pod), C.byref(info)))
    return dict(closed_room=bool(info.closedRoom), grid=bool(info.grid), shadow_lists=bool(info.shadowLists), room=np.array(info.room[:], np.float32),
                grid_center=np.array(info.gridCenter[:], np.float32), grid_reach=float(info.gridReach))


def stage_shadow_lists(scene: Scene):
    """Host-only: the per-light visibility lists create() stages for a large scene (device_scene.h DShadowLists), or None when
    the scene gets none. -> dict(n=bins per cube-face axis, lights=[sphere index], start=[nLights * 6 n^2 + 1], key, index)."""
    L = capi.lib()
    pod = scene.pod()
    n, nl = C.c_int32(), C.c_int32()
    items = L.kajo_hip_stage_shadow_lists(C.byref(pod), C.byref(n), C.byref(nl), None, None, 0, None, None, 0)
    if items < 0:
        capi.check(items)
    if n.value == 0:
        return None
    lights = np.zeros(nl.value, np.int32)
    start = np.zeros(nl.value * 6 * n.value * n.value + 1, np.uint32)
    key = np.zeros(items, np.float32)
    index = np.zeros(items, np.uint32)
    rc = L.kajo_hip_stage_shadow_lists(C.byref(pod), C.byref(n), C.byref(nl), lights.ctypes.data_as(C.c_void_p), start.ctypes.data_as(C.c_void_p),
                                       start.size, key.ctypes.data_as(C.c_void_p), index.ctypes.data_as(C.c_void_p), items)
    if rc < 0:
        capi.check(rc)
    return dict(n=n.value, lights=lights, start=start, key=key, index=index)
