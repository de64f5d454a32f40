"""Seeded scenes generated where they are used (tests/test_hip_whole_frames.py; tools/whole_frame.py, oracle_vs_reference.py): not fixtures."""
import numpy as np
from kajo_amd.scene import Scene


def rotation(rng):
    """a rigid rotation (Rodrigues), as a 4 x 4"""
    ax = rng.normal(size=3)
    ax /= np.linalg.norm(ax)
    a = rng.uniform(0, 2 * np.pi)
    K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
    R = np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)
    M = np.eye(4)
    M[:3, :3] = R
    return M


def mixed_scene(base, seed):
    from kajo_amd.scene import material, sphere_record, plane_record
    rng = np.random.default_rng(seed)
    lin = lambda c: np.float32(c) ** np.float32(2.2)
    col = lambda lo=.1, hi=.9: [lin(rng.uniform(lo, hi)) for _ in range(3)]

    def some_material():
        k = rng.integers(5)
        if k == 0:
            return material(diffuse=col())
        if k == 1:
            return material(specular=col(), exponent=float(rng.choice([2, 10, 50, 100, 400])))
        if k == 2:
            return material(specular=col(.5, .9), exponent=0.0)  # ideal reflector
        if k == 3:  # (glass over an ideal reflector, exponent 0, is NaN over whole regions in the reference: nothing to compare there)
            return material(specular=col(.2, .6), transparency=col(.7, 1.0), exponent=float(rng.choice([20, 100])), ior=float(rng.choice([1.1, 1.5, 2.0, 2.4])))
        return material(diffuse=col(.1, .5), specular=col(.1, .5), exponent=float(rng.choice([5, 30])))
    planes = base.planes.copy()
    for i in range(len(planes)):
        planes[i, 16:38] = some_material() if rng.random() < .7 else planes[i, 16:38]
    recs = []
    for _ in range(int(rng.integers(8, 21))):
        T = np.eye(4)
        T[:3, 3] = (rng.uniform(-4, 8), rng.uniform(-1.5, .6), rng.uniform(-1.5, 4))
        M = (T @ rotation(rng)).astype(np.float32)
        recs.append(sphere_record(M.T.reshape(16).copy(), some_material(), float(rng.uniform(.2, 1.1))))
    for _ in range(int(rng.integers(1, 5))):
        T = np.eye(4)
        T[:3, 3] = (rng.uniform(-4, 8), rng.uniform(-1.8, -.8), rng.uniform(-1, 4))
        M = (T @ rotation(rng)).astype(np.float32)
        recs.append(sphere_record(M.T.reshape(16).copy(), material(emission=[lin(rng.uniform(6, 16))] * 3), float(rng.uniform(.1, .5))))
    return Scene(base.background, base.view, base.proj, np.stack(recs), planes, "mix%d" % seed)

