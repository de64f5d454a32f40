"""Where to look in a full-size frame: 64 x 32-pixel crops on the features of a scene that exercise different
parts of the integrator (glass, mirror, Phong lobe, emitter, shadow edges) plus the corners of the image.
Used by tests/test_hip_workloads.py and tests/golden/make_golden_frames2.py; geometry only, no rendering."""
import numpy as np

CW, CH = 64, 32


def project(scene, P, W, H):
    """World point -> pixel (x, y), row 0 = top (Renderer.cpp:56-57: sy = (H - y) / H)."""
    view = scene.view.reshape(4, 4).T.astype(np.float64)  # column-major flat -> matrix
    proj = scene.proj.reshape(4, 4).T.astype(np.float64)
    c = proj @ view @ np.array([P[0], P[1], P[2], 1.0])
    n = c[:3] / c[3]
    return (n[0] * .5 + .5) * W, (1 - (n[1] * .5 + .5)) * H


def _clamp(x, y, W, H, w=CW, h=CH):
    w, h = min(w, W), min(h, H)
    return int(min(max(x - w // 2, 0), W - w)), int(min(max(y - h // 2, 0), H - h)), w, h


def feature_crops(scene, W, H, max_spheres=6):
    """[(name, x0, y0, w, h)]: sphere centres (first the ones with a transparent / specular / emissive material), the
    right-hand silhouette of the first sphere, the floor under it (its shadow), the four corners, the image centre."""
    crops = []
    sph = scene.spheres
    order = list(range(min(scene.n_spheres, 64)))
    # material floats of a sphere record: [16:38] = ambient, diffuse, specular, emission, transparency (4 each), exponent, ior
    def score(i):
        m = sph[i, 16:38]
        return -(4 * (m[16:19].sum() > 0) + 3 * (m[12:15].sum() > 0) + 2 * (m[8:11].sum() > 0 and m[20] != 0) + (m[8:11].sum() > 0))
    order.sort(key=score)
    seen = []
    for i in order:
        c, r = sph[i, 12:15], sph[i, 38]
        x, y = project(scene, c, W, H)
        if not (0 <= x < W and 0 <= y < H):
            continue
        if any(abs(x - sx) < CW and abs(y - sy) < CH for sx, sy in seen):
            continue
        seen.append((x, y))
        crops.append(("sphere %d centre" % i, *_clamp(x, y, W, H)))
        if len(seen) == 1:
            xe, _ = project(scene, c + np.array([0, 0, r], np.float32), W, H)
            xr, yr = project(scene, c + np.array([r, 0, 0], np.float32), W, H)
            ex = x + max(abs(xe - x), abs(xr - x))  # a silhouette point to the right of the centre (approximately)
            crops.append(("sphere %d silhouette" % i, *_clamp(ex, y, W, H)))
            xf, yf = project(scene, np.array([c[0], 1.0, c[2]]) + np.array([r * .9, 0, 0]), W, H)  # the floor is y = 1
            if 0 <= xf < W and 0 <= yf < H:
                crops.append(("floor under sphere %d" % i, *_clamp(xf, yf, W, H)))
        if len(seen) >= max_spheres:
            break
    for name, x, y in (("top left", 0, 0), ("bottom right", W, H), ("top right", W, 0), ("bottom left", 0, H), ("centre", W / 2, H / 2)):
        crops.append((name, *_clamp(x, y, W, H)))
    # de-duplicate rectangles
    out, rects = [], set()
    for c in crops:
        if c[1:] not in rects:
            rects.add(c[1:])
            out.append(c)
    return out


def mirror_crop(scene, W, H):
    """A crop where camera rays land on a plane with an ideal-mirror material (specular colour, exponent 0). Located with the
    oracle's closest-hit walk on a coarse grid of camera rays (test infrastructure)."""
    from oraclelib import OracleLib
    pl = scene.planes
    mirrors = [i + 1 for i in range(scene.n_planes) if pl[i, 16 + 8:16 + 11].sum() > 0 and pl[i, 16 + 20] == 0 and pl[i, 16 + 4:16 + 7].sum() == 0]
    if not mirrors:
        return None
    h = OracleLib("oracle").create(scene, 0)
    p1, p2, p3, origin = h.camera_basis().astype(np.float64)
    gx, gy = np.meshgrid((np.arange(48) + .5) / 48, (np.arange(27) + .5) / 27)
    d = p1 + gx.reshape(-1, 1) * (p2 - p1) + (1 - gy.reshape(-1, 1)) * (p3 - p1) - origin
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    idx = h.trace(np.repeat(origin[None], len(d), 0), d)["idx"].reshape(27, 48)
    ys, xs = np.nonzero(np.isin(idx, mirrors))
    if len(xs) == 0:
        return None
    k = len(xs) // 2
    return ("mirror wall", *_clamp((xs[k] + .5) / 48 * W, (ys[k] + .5) / 27 * H, W, H))


def crops_for(scene, W, H, limit=10):
    """The crops the workload tests compare: feature crops with the mirror-wall crop in fourth place."""
    crops = feature_crops(scene, W, H)
    m = mirror_crop(scene, W, H)
    if m:
        crops.insert(3, m)
    return crops[:limit]
