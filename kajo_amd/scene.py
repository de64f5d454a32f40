"""Flat scene model handed across the C ABI (ctypes mirror of include/kajo_scene.h).

Field for field this is the reference's scene::Scene (scene/Scene.h:11-62): matrices are
column-major glm::mat4 images, colours are linear RGBA as scene::Parser leaves them
(scene/Parser.cpp:70-92). The module also holds the synthetic scene builders for the
benchmark configurations that are not shipped as JSON (C4, C5).
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass, field

import numpy as np


class KajoMaterial(C.Structure):
    _fields_ = [
        ("ambient", C.c_float * 4),
        ("diffuse", C.c_float * 4),
        ("specular", C.c_float * 4),
        ("emission", C.c_float * 4),
        ("transparency", C.c_float * 4),
        ("specularExponent", C.c_float),
        ("refractiveIndex", C.c_float),
    ]


class KajoSphere(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("material", KajoMaterial), ("radius", C.c_float)]


class KajoPlane(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("material", KajoMaterial)]


class KajoCamera(C.Structure):
    _fields_ = [("transform", C.c_float * 16), ("projection", C.c_float * 16)]


class KajoScene(C.Structure):
    _fields_ = [
        ("backgroundColor", C.c_float * 4),
        ("camera", KajoCamera),
        ("nSpheres", C.c_int32),
        ("nPlanes", C.c_int32),
        ("spheres", C.POINTER(KajoSphere)),
        ("planes", C.POINTER(KajoPlane)),
    ]


MATERIAL_FLOATS = 22
SPHERE_FLOATS = 39
PLANE_FLOATS = 38
assert C.sizeof(KajoMaterial) == 4 * MATERIAL_FLOATS
assert C.sizeof(KajoSphere) == 4 * SPHERE_FLOATS
assert C.sizeof(KajoPlane) == 4 * PLANE_FLOATS

# offsets inside the 22-float material image
_AMBIENT, _DIFFUSE, _SPECULAR, _EMISSION, _TRANSPARENCY, _EXPONENT, _IOR = 0, 4, 8, 12, 16, 20, 21


def material(diffuse=None, specular=None, emission=None, transparency=None, exponent=0.0, ior=1.0):
    """22-float material image; unset colours are (0,0,0,0) and set ones carry alpha 1,
    exactly as scene::Material() + Parser::parseColor leave them (scene/Scene.cpp:10-14,
    scene/Parser.cpp:75-92)."""
    m = np.zeros(MATERIAL_FLOATS, np.float32)
    for off, c in ((_DIFFUSE, diffuse), (_SPECULAR, specular), (_EMISSION, emission), (_TRANSPARENCY, transparency)):
        if c is not None:
            c = np.asarray(c, np.float32).ravel()
            m[off:off + 3] = c[:3]
            m[off + 3] = c[3] if c.size > 3 else 1.0
    m[_EXPONENT] = exponent
    m[_IOR] = ior
    return m


def translate(x, y, z):
    """glm::translate(mat4(1), v): identity with the offset in column 3."""
    m = np.eye(4, dtype=np.float32)
    m[:3, 3] = (x, y, z)
    return m.T.reshape(16).copy()  # column-major image


@dataclass
class Scene:
    """Host-side scene: numpy images of the POD arrays."""

    background: np.ndarray  # (4,)
    view: np.ndarray  # (16,) column-major
    proj: np.ndarray  # (16,)
    spheres: np.ndarray  # (nSpheres, 39)
    planes: np.ndarray  # (nPlanes, 38)
    name: str = "scene"
    _keep: list = field(default_factory=list, repr=False)

    def __post_init__(self):
        self.background = np.ascontiguousarray(self.background, np.float32).reshape(4)
        self.view = np.ascontiguousarray(self.view, np.float32).reshape(16)
        self.proj = np.ascontiguousarray(self.proj, np.float32).reshape(16)
        self.spheres = np.ascontiguousarray(self.spheres, np.float32).reshape(-1, SPHERE_FLOATS)
        self.planes = np.ascontiguousarray(self.planes, np.float32).reshape(-1, PLANE_FLOATS)

    @property
    def n_spheres(self):
        return self.spheres.shape[0]

    @property
    def n_planes(self):
        return self.planes.shape[0]

    @property
    def n_lights(self):
        # emission != vec4(0) (renderer/cpu/Shader.cpp:57), spheres only
        return int(np.any(self.spheres[:, 16 + _EMISSION:16 + _EMISSION + 4] != 0, axis=1).sum())

    def pod(self) -> KajoScene:
        """ctypes KajoScene pointing into this object's arrays (kept alive by self)."""
        s = KajoScene()
        C.memmove(s.backgroundColor, self.background.ctypes.data, 16)
        C.memmove(s.camera.transform, self.view.ctypes.data, 64)
        C.memmove(s.camera.projection, self.proj.ctypes.data, 64)
        s.nSpheres = self.n_spheres
        s.nPlanes = self.n_planes
        s.spheres = C.cast(self.spheres.ctypes.data, C.POINTER(KajoSphere))
        s.planes = C.cast(self.planes.ctypes.data, C.POINTER(KajoPlane))
        return s

    def write_pod(self, path: str) -> None:
        """The flat binary image kajo_render --scene-pod reads (kajo_amd/host/Main.cpp): int32 nSpheres, nPlanes; background;
        view; projection; sphere records (39 f32); plane records (38 f32) -- scene::Scene's own record layout."""
        with open(path, "wb") as f:
            f.write(np.array([self.n_spheres, self.n_planes], np.int32).tobytes())
            for a in (self.background, self.view, self.proj, self.spheres, self.planes):
                f.write(np.ascontiguousarray(a, np.float32).tobytes())

    def with_aspect(self, aspect: float) -> "Scene":
        """Same scene, projection rebuilt for another aspect ratio. glm::perspective
        (gtc/matrix_transform.inl:223-245) only puts the aspect into element [0][0]:
        [0][0] = (2 near) / (right - left) with right = range * aspect, so
        new[0][0] = [1][1] / aspect up to rounding. Used for synthetic frames only; parity
        fixtures carry projections produced by the reference parser itself."""
        proj = self.proj.copy()
        proj[0] = np.float32(proj[5] / np.float32(aspect))
        return Scene(self.background, self.view, proj, self.spheres, self.planes, self.name)

    def to_npz_dict(self, prefix=""):
        return {
            prefix + "background": self.background,
            prefix + "view": self.view,
            prefix + "proj": self.proj,
            prefix + "spheres": self.spheres,
            prefix + "planes": self.planes,
        }

    @staticmethod
    def from_npz(z, prefix="", name="scene") -> "Scene":
        return Scene(z[prefix + "background"], z[prefix + "view"], z[prefix + "proj"], z[prefix + "spheres"],
                     z[prefix + "planes"], name)


def sphere_record(transform16, mat22, radius):
    r = np.zeros(SPHERE_FLOATS, np.float32)
    r[:16] = transform16
    r[16:38] = mat22
    r[38] = radius
    return r


def plane_record(transform16, mat22):
    r = np.zeros(PLANE_FLOATS, np.float32)
    r[:16] = transform16
    r[16:38] = mat22
    return r


def caustics_scene(base: Scene) -> Scene:
    """C4: the spheres.json room (its six planes incl. the ideal-reflector
    wall 1, data/spheres.json:42-79) with Phong spheres, the glass sphere and THREE emissive
    spheres; authored by this repo."""
    lin = lambda c: np.float32(c) ** np.float32(2.2)  # Parser.cpp:70-73
    spheres = [
        sphere_record(translate(-2, 0, 0), material(specular=[lin(8 / 15)] * 3, transparency=[lin(14 / 15), lin(14 / 15), 1.0],
                                                     exponent=100, ior=2.0), 1.0),
        sphere_record(translate(1, 0, .5), material(specular=[lin(10 / 15), lin(2 / 15), lin(2 / 15)], exponent=100), 1.0),
        sphere_record(translate(4, 0, 1), material(specular=[lin(2 / 15), lin(10 / 15), lin(2 / 15)], exponent=20), 1.0),
        sphere_record(translate(7, 0, 1.5), material(diffuse=[lin(2 / 15), lin(2 / 15), lin(10 / 15)]), 1.0),
        sphere_record(translate(-1, -1.5, 2), material(emission=[lin(16.0)] * 3), .3),
        sphere_record(translate(3, -1.6, -.5), material(emission=[lin(12.0), lin(12.0), lin(8.0)]), .25),
        sphere_record(translate(6, -1.4, 3), material(emission=[lin(8.0), lin(10.0), lin(14.0)]), .25),
    ]
    return Scene(base.background, base.view, base.proj, np.stack(spheres), base.planes, "caustics3")


def stress_scene(base: Scene, n_spheres=1000, n_lights=16, seed=1234) -> Scene:
    """C5: the six room planes + `n_spheres` rigid (translate-only) spheres
    r in [0.05, 0.15] on a jittered grid inside the room, materials cycling diffuse / Phong,
    plus `n_lights` emissive spheres r = 0.1 with emission rgb(16,16,16); seeded."""
    rng = np.random.default_rng(seed)
    lin = lambda c: np.float32(c) ** np.float32(2.2)
    # room interior (world is Y-down): x in [-8, 10], y in [-2, 1], z in [-2, 6]
    nx, ny, nz = 20, 5, 10
    assert nx * ny * nz >= n_spheres
    cells = rng.permutation(nx * ny * nz)[:n_spheres]
    recs = []
    for k, cidx in enumerate(cells):
        ix, iy, iz = cidx % nx, (cidx // nx) % ny, cidx // (nx * ny)
        j = rng.random(3) * .5 + .25
        x = -3.0 + (ix + j[0]) * (12.0 / nx)
        y = -1.8 + (iy + j[1]) * (2.6 / ny)
        z = -1.5 + (iz + j[2]) * (6.0 / nz)
        r = .05 + .1 * rng.random()
        col = [lin(.2 + .6 * rng.random()) for _ in range(3)]
        if k % 2 == 0:
            m = material(diffuse=col)
        else:
            m = material(specular=col, exponent=float(rng.choice([10, 50, 100])))
        recs.append(sphere_record(translate(x, y, z), m, r))
    for k in range(n_lights):
        x = -2.5 + 11.0 * (k % 8 + .5) / 8
        z = 0.0 + 4.0 * (k // 8)
        recs.append(sphere_record(translate(x, -1.85, z), material(emission=[lin(16.0)] * 3), .1))
    return Scene(base.background, base.view, base.proj, np.stack(recs), base.planes, "stress%d_%d" % (n_spheres, n_lights))
