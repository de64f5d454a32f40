"""The host side under AddressSanitizer + UBSan (SURVEY.md section 5 "race detection / sanitizers"; CPU only: GPU sanitizers are not
available on the pool). tools/host_san.cpp is compiled from the PRODUCT's own sources -- kajo_amd/csrc/stage.cpp (object records, grid,
visibility lists), kajo_amd/host/scene/SceneLoader.cpp (Kajo's JSON dialect), kajo_amd/csrc/launch_order.h (cost order, parted launch
tail) and render_args.h (tile slots, side-buffer slots) -- with -fsanitize=address,undefined and run on the tests' scenes; the launch order
and the side-buffer slots are also checked against a model written here."""
import os
import shutil
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def san(tmp_path_factory):
    if shutil.which("g++") is None:
        pytest.skip("g++ not available")
    d = tmp_path_factory.mktemp("san")
    exe = str(d / "host_san")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer",
           "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "kajo_amd", "csrc"),
           "-I" + os.path.join(ROOT, "kajo_amd", "host"), os.path.join(ROOT, "tools", "host_san.cpp"),
           os.path.join(ROOT, "kajo_amd", "csrc", "stage.cpp"), os.path.join(ROOT, "kajo_amd", "host", "scene", "SceneLoader.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]

    def run(*args, timeout=600):
        p = subprocess.run([exe, *map(str, args)], capture_output=True, text=True, timeout=timeout,
                           env=dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1"))
        assert p.returncode == 0, "host_san %s -> %d\n%s\n%s" % (" ".join(map(str, args)), p.returncode, p.stdout[-2000:], p.stderr[-4000:])
        assert "runtime error" not in p.stderr and "AddressSanitizer" not in p.stderr, p.stderr[-4000:]
        return p.stdout

    run.dir = d
    return run


def test_scene_staging(san, scenes):
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from kajo_amd.scene import stress_scene
    from test_shadow_lists_cpu import adversarial_scene
    a = scenes["spheres_a169"]
    files = []
    for name, sc in [("spheres", a), ("dialect", scenes["dialect_a1"]), ("caustics", scenes["caustics_a169"]), ("test", scenes["test_a1"]),
                     ("stress1000", stress_scene(a, 1000, 16)), ("stress300", stress_scene(a, 300, 8, seed=77)), ("stress120", stress_scene(a, 120, 1, seed=3))] + \
                    [("adv%d" % s, adversarial_scene(a, s)) for s in (1, 2, 3, 4)]:
        path = str(san.dir / (name + ".pod"))
        sc.write_pod(path)
        files.append(path)
    out = san("stage", *files)
    assert out.count(": ok") == 2 * len(files)


def test_scene_loader(san):
    files = [os.path.join(ROOT, "kajo_amd", "data", n) for n in ("caustics.json", "dialect.json")]
    files += [p for p in (os.path.join("/root/reference/data", n) for n in ("spheres.json", "test.json")) if os.path.exists(p)]
    for aspect in (1.0, 1920.0 / 1080.0, 640.0 / 480.0):  # (the aspects of the six parsed scenes of tests/golden/scenes.npz)
        out = san("parse", aspect, *files)
        assert out.count(" spheres ") == len(files)


def model_order(trips, n, w, slots, parts):
    """launch_order.h restated: -> (nParted, order words)"""
    cost = trips.reshape(n, w).max(1) if n else np.zeros(0, np.uint32)
    plain = np.argsort(-cost.astype(np.int64), kind="stable").astype(np.uint32)
    per = slots // w
    n_parted = 0 if (n >= (1 << 28) or n < 2 * per) else min(n // 2, per * 4 // 8)
    if parts < 2 or n_parted == 0:
        return n_parted, plain
    head, tail = plain[:n - n_parted], plain[n - n_parted:]
    words = (tail[:, None] | (np.arange(parts, dtype=np.uint32)[None, :] << 28) | np.uint32(0x80000000)).reshape(-1)
    return n_parted, np.concatenate([head, words]).astype(np.uint32)


@pytest.mark.parametrize("n,w,parts", [(0, 1, 4), (1, 1, 4), (5000, 1, 4), (32400, 1, 4), (32400, 1, 2), (32400, 1, 8), (8100, 4, 3), (129600, 1, 7),
                                       (1 << 22, 1, 8)])
def test_launch_order_and_side_slots(san, n, w, parts):
    rng = np.random.default_rng(n + 7 * w + parts)
    trips = rng.integers(1, 2000, n * w, dtype=np.uint32)
    if n > 100:
        trips[:50] = trips[50]  # ties: equal costs keep image order
    slots, threads = 5120, 64 * w
    src, dst = str(san.dir / "order_in.bin"), str(san.dir / "order_out.bin")
    with open(src, "wb") as f:
        np.array([n, w, slots, parts, threads], np.uint32).tofile(f)
        trips.tofile(f)
    san("order", src, dst)  # (checks inside: every later part's side-buffer slots in range, none taken twice)
    got = np.fromfile(dst, np.uint32)
    n_parted, words = model_order(trips, n, w, slots, parts)
    assert got[0] == n_parted and got[1] == words.size
    assert np.array_equal(got[2:], words)
    if n_parted and parts >= 2:
        # what the fold kernel assumes (aux_kernels.hip): the j-th parted block's parts are physical workgroups partedFirst + j * parts + k,
        # and part k > 0 of it owns slots [(k - 1) * sideStride + j * threads, + threads) of the side buffers
        first = n - n_parted
        phys = np.arange(first, words.size, dtype=np.uint64)
        k = (words[first:] >> 28) & 7
        assert np.array_equal(k, (phys - first) % parts) and (words[first:] >> 31).all() and not (words[:first] >> 28).any()
        j = (phys - first) // parts
        assert np.array_equal(words[first:] & 0x0fffffff, words[first + j * parts] & 0x0fffffff)
        side = (k.astype(np.uint64) - 1) * (n_parted * threads) + j * threads
        assert side[k > 0].max() + threads <= (parts - 1) * n_parted * threads


def test_launch_order_through_the_c_abi():
    """kajo_hip_launch_order (host-only entry point of libkajo_hip.so) returns the same words."""
    import ctypes as C
    from kajo_amd import capi
    L = capi.lib()
    n, w, parts = 20000, 1, 4
    trips = np.random.default_rng(3).integers(1, 500, n * w, dtype=np.uint32)
    n_parted = C.c_uint32()
    need = L.kajo_hip_launch_order(trips.ctypes.data_as(C.c_void_p), n, w, 5120, parts, None, 0, C.byref(n_parted))
    want_parted, words = model_order(trips, n, w, 5120, parts)
    assert need == words.size and n_parted.value == want_parted == 2560
    out = np.zeros(need, np.uint32)
    assert L.kajo_hip_launch_order(trips.ctypes.data_as(C.c_void_p), n, w, 5120, parts, out.ctypes.data_as(C.c_void_p), need, None) == need
    assert np.array_equal(out, words)
    assert L.kajo_hip_launch_order(trips.ctypes.data_as(C.c_void_p), n, w, 5120, parts, out.ctypes.data_as(C.c_void_p), need - 1, None) == capi.KAJO_E_INVALID
    assert L.kajo_hip_launch_order(trips.ctypes.data_as(C.c_void_p), 1 << 28, w, 5120, parts, None, 0, None) == capi.KAJO_E_INVALID


@pytest.mark.parametrize("w,h,tile,owners", [(1920, 1080, (64, 16), 1), (1920, 1080, (64, 16), 8), (200, 70, (32, 8), 3), (3840, 2160, (64, 16), 8),
                                             (1, 1, (64, 16), 2), (8191, 33, (8, 32), 5)])
def test_tile_slots(san, w, h, tile, owners):
    out = san("tiles", w, h, tile[0], tile[1], owners)
    from kajo_amd.tiles import TileLayout
    assert "%d slots per owner" % TileLayout(w, h, owners, tile).slots_per_owner in out
