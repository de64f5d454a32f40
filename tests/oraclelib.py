"""ctypes bindings of the parity checkers. TEST INFRASTRUCTURE ONLY.

  OracleLib("oracle")      -> oracle/libkajo_oracle.so  (this repo's CPU restatement)
  OracleLib("oracle_fast") -> oracle/libkajo_oracle_fast.so (the same, reference flag set; CPU-baseline timing only)
  OracleLib("ref")         -> oracle/_ref/libkajo_ref.so        (the compiled reference, fast flags)
  OracleLib("ref_strict")  -> oracle/_ref/libkajo_ref_strict.so (the compiled reference, -O2)

The two reference builds and the restatement export the same entry points under the prefixes
kref_ / koracle_, so one fixture can be run through any of them.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from kajo_amd.scene import KajoPlane, KajoScene, KajoSphere, Scene

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATHS = {
    "oracle": os.path.join(ROOT, "oracle", "libkajo_oracle.so"),
    "oracle_fast": os.path.join(ROOT, "oracle", "libkajo_oracle_fast.so"),  # reference flag set: timing only
    "ref": os.path.join(ROOT, "oracle", "_ref", "libkajo_ref.so"),
    "ref_strict": os.path.join(ROOT, "oracle", "_ref", "libkajo_ref_strict.so"),
}


def available(which: str) -> bool:
    return os.path.exists(PATHS[which])


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def _f32(a):
    return np.ascontiguousarray(a, np.float32)


class OracleLib:
    def __init__(self, which="oracle"):
        self.which = which
        self.is_ref = which.startswith("ref")
        self.prefix = "kref_" if self.is_ref else "koracle_"
        self.lib = C.CDLL(PATHS[which])
        self._fn("create").restype = C.c_void_p
        if self.is_ref:
            self.lib.kref_create_from_file.restype = C.c_void_p
            self.lib.kref_create_from_file.argtypes = [C.c_char_p, C.c_float]
            self.lib.kref_render_native.restype = C.c_double
        else:
            self.lib.koracle_render.restype = C.c_double
            self.lib.koracle_render_native.restype = C.c_double

    def _fn(self, name):
        return getattr(self.lib, self.prefix + name)

    # -- handles -------------------------------------------------------------------
    def create(self, scene: Scene, math=0):
        pod = scene.pod()
        if self.is_ref:
            h = self._fn("create")(C.byref(pod))
        else:
            h = self._fn("create")(C.byref(pod), C.c_int(math))
        return Handle(self, h, scene, math)

    def create_from_file(self, path: str, aspect: float):
        assert self.is_ref
        h = self.lib.kref_create_from_file(path.encode(), C.c_float(aspect))
        if not h:
            raise RuntimeError("reference parser failed on " + path)
        return Handle(self, h, None, 0)

    # -- stateless -----------------------------------------------------------------
    def rng_from_seed(self, seed, n):
        out = np.zeros((n, 4), np.float32)
        st = np.zeros(2, np.uint64)
        self._fn("rng_from_seed")(C.c_uint(seed), C.c_int(n), _p(out), _p(st))
        return out, st

    def rng_from_state(self, state, n):
        state = np.ascontiguousarray(state, np.uint64)
        out = np.zeros((n, 4), np.float32)
        st = np.zeros(2, np.uint64)
        self._fn("rng_from_state")(_p(state), C.c_int(n), _p(out), _p(st))
        return out, st

    def flip_coin(self, state, p):
        state = np.ascontiguousarray(state, np.uint64)
        v = C.c_int()
        pr = C.c_float()
        self._fn("flip_coin")(_p(state), C.c_float(p), C.byref(v), C.byref(pr))
        return bool(v.value), pr.value

    def resolve(self, accum, npass, math=0):
        accum = _f32(accum).reshape(-1, 4)
        px = np.zeros(accum.shape[0], np.uint32)
        if self.is_ref:
            self.lib.kref_resolve(C.c_int(accum.shape[0]), _p(accum), C.c_int(npass), _p(px))
        else:
            self.lib.koracle_resolve(C.c_int(math), C.c_int(accum.shape[0]), _p(accum), C.c_int(npass), _p(px))
        return px


class Handle:
    def __init__(self, lib: OracleLib, h, scene, math):
        self.L = lib
        self.h = C.c_void_p(h)
        self.scene = scene
        self.math = math

    def close(self):
        if self.h:
            self.L._fn("destroy")(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def export_scene(self, name="scene") -> Scene:
        """Reference handles only: the scene held by the handle as POD arrays."""
        ns, npl = C.c_int(), C.c_int()
        self.L.lib.kref_counts(self.h, C.byref(ns), C.byref(npl))
        bg = np.zeros(4, np.float32)
        view = np.zeros(16, np.float32)
        proj = np.zeros(16, np.float32)
        sph = np.zeros((ns.value, 39), np.float32)
        pl = np.zeros((npl.value, 38), np.float32)
        self.L.lib.kref_export(self.h, _p(bg), _p(view), _p(proj), _p(sph), _p(pl))
        return Scene(bg, view, proj, sph, pl, name)

    def staged(self, n_objects):
        out = np.zeros((n_objects, 17), np.float32)
        self.L._fn("staged")(self.h, _p(out))
        return out

    def camera_basis(self):
        out = np.zeros(12, np.float32)
        self.L._fn("camera_basis")(self.h, _p(out))
        return out.reshape(4, 3)

    def trace(self, origins, dirs):
        origins, dirs = _f32(origins), _f32(dirs)
        n = origins.shape[0]
        idx = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        pos, nor, tan, bin_ = (np.zeros((n, 3), np.float32) for _ in range(4))
        self.L._fn("trace")(self.h, C.c_int(n), _p(origins), _p(dirs), _p(idx), _p(t), _p(pos), _p(nor), _p(tan), _p(bin_))
        return dict(idx=idx, t=t, position=pos, normal=nor, tangent=tan, binormal=bin_)

    def sample(self, kind, origins, dirs, states, color, param=0.0, light_sphere=0):
        origins, dirs = _f32(origins), _f32(dirs)
        states = np.ascontiguousarray(states, np.uint64)
        n = origins.shape[0]
        color = _f32(color)
        hit = np.zeros(n, np.int32)
        d = np.zeros((n, 3), np.float32)
        pdf = np.zeros(n, np.float32)
        f = np.zeros((n, 3), np.float32)
        pq = np.zeros(n, np.float32)
        fin = np.zeros((n, 2), np.uint64)
        self.L._fn("sample")(self.h, C.c_int(kind), C.c_int(n), _p(origins), _p(dirs), _p(states), _p(color),
                             C.c_float(param), C.c_int(light_sphere), _p(hit), _p(d), _p(pdf), _p(f), _p(pq), _p(fin))
        return dict(hit=hit, dir=d, pdf=pdf, f=f, pq=pq, final=fin)

    def shade(self, origins, dirs, states, depth_limit=8):
        origins, dirs = _f32(origins), _f32(dirs)
        states = np.ascontiguousarray(states, np.uint64)
        n = origins.shape[0]
        rgb = np.zeros((n, 3), np.float32)
        fin = np.zeros((n, 2), np.uint64)
        if self.L.is_ref:
            self.L.lib.kref_set_depth_limit(C.c_int(depth_limit))
            self.L.lib.kref_shade(self.h, C.c_int(n), _p(origins), _p(dirs), _p(states), _p(rgb), _p(fin))
            self.L.lib.kref_set_depth_limit(C.c_int(8))
        else:
            self.L.lib.koracle_shade(self.h, C.c_int(n), _p(origins), _p(dirs), _p(states), C.c_int(depth_limit),
                                     _p(rgb), _p(fin))
        return rgb, fin

    def render(self, W, H, S=32, passes=1, seed=236367, depth_limit=8, first_pass=1, rect=None, accum=None,
               threads=None, counters=False):
        """Per-sample-stream frame -> (H, W, 4) float32 accumulation (sum over passes of
        radiance / S, NOT divided by the pass count)."""
        if accum is None:
            accum = np.zeros((H, W, 4), np.float32)
        x0, y0, w, h = rect if rect else (0, 0, W, H)
        ctr = np.zeros(4, np.uint64)
        if self.L.is_ref:
            self.L.lib.kref_set_depth_limit(C.c_int(depth_limit))
            self.L.lib.kref_render(self.h, C.c_int(W), C.c_int(H), C.c_int(S), C.c_int(first_pass), C.c_int(passes),
                                   C.c_uint64(seed), C.c_int(x0), C.c_int(y0), C.c_int(w), C.c_int(h), _p(accum))
            self.L.lib.kref_set_depth_limit(C.c_int(8))
        else:
            if threads is None:
                threads = min(8, os.cpu_count() or 1)
            self.L.lib.koracle_render(self.h, C.c_int(W), C.c_int(H), C.c_int(S), C.c_int(first_pass), C.c_int(passes),
                                      C.c_uint64(seed), C.c_int(depth_limit), C.c_int(x0), C.c_int(y0), C.c_int(w),
                                      C.c_int(h), _p(accum), C.c_int(threads), _p(ctr) if counters else None)
        return (accum, ctr) if counters else accum

    def render_native(self, W, H, passes, threads, depth_limit=8):
        """The reference's own stream discipline + threading; returns (seconds, image)."""
        if self.L.is_ref:
            px = np.zeros(W * H, np.uint32)
            s = self.L.lib.kref_render_native(self.h, C.c_int(W), C.c_int(H), C.c_int(passes), C.c_int(threads), _p(px))
            return s, px.reshape(H, W)
        acc = np.zeros((H, W, 4), np.float32)
        s = self.L.lib.koracle_render_native(self.h, C.c_int(W), C.c_int(H), C.c_int(passes), C.c_int(threads),
                                             C.c_int(depth_limit), _p(acc))
        return s, acc


def camera_ray(handle, W, H, S, x, y, sample, npass=1, seed=236367):
    """Oracle only: camera ray (origin, direction) of one path and the generator state after its jitter draw --
    what shade() / kajo_hip_kat_shade take, so that the paths of a frame can be replayed one by one."""
    ray = np.zeros(6, np.float32)
    state = np.zeros(2, np.uint64)
    handle.L.lib.koracle_camera_ray(handle.h, C.c_int(W), C.c_int(H), C.c_int(S), C.c_int(npass), C.c_uint64(seed), C.c_int(x),
                                    C.c_int(y), C.c_int(sample), _p(ray), _p(state))
    return ray[:3].copy(), ray[3:].copy(), state


def debug_path(handle, W, H, S, x, y, sample, npass=1, seed=236367, depth_limit=8):
    """Oracle only: event log of one camera path (see koracle_debug_path)."""
    out = np.zeros(4096, np.float32)
    rgb = np.zeros(3, np.float32)
    n = handle.L.lib.koracle_debug_path(handle.h, C.c_int(W), C.c_int(H), C.c_int(S), C.c_int(npass), C.c_uint64(seed),
                                        C.c_int(depth_limit), C.c_int(x), C.c_int(y), C.c_int(sample), _p(out),
                                        C.c_int(out.size), _p(rgb))
    return out[:n].reshape(-1, 4), rgb
