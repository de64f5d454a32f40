"""CPU-side checks of the C ABI library: it loads, exports exactly what include/kajo_hip.h
declares, stages scenes like the oracle, validates arguments and refuses to run without a GPU
(there is no CPU fallback in the product)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from kajo_amd import capi
from kajo_amd.renderer import HipRenderer, stage_scene
from oraclelib import OracleLib, available

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_header_symbols():
    L = capi.lib()
    header = open(os.path.join(ROOT, "include", "kajo_hip.h")).read()
    declared = set(re.findall(r"\b(kajo_hip_[a-z0-9_]+)\s*\(", header))  # kajo_hip_debug_profile is not part of the ABI
    assert declared == set(capi.EXPORTS), declared ^ set(capi.EXPORTS)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.kajo_hip_version()


def test_pod_layout_matches_header():
    # sizes the C side assumes (include/kajo_scene.h comments)
    from kajo_amd import scene as S
    assert C.sizeof(S.KajoMaterial) == 88 and C.sizeof(S.KajoSphere) == 156 and C.sizeof(S.KajoPlane) == 152
    assert C.sizeof(S.KajoCamera) == 128
    assert C.sizeof(capi.KajoParams) == 48 and C.sizeof(capi.KajoCounters) == 80


@pytest.mark.skipif(not available("oracle"), reason="oracle not built")
def test_staging_bit_exact_with_oracle(scenes):
    O = OracleLib("oracle")
    for key, sc in scenes.items():
        inv, basis = stage_scene(sc)
        h = O.create(sc)
        assert np.array_equal(inv, h.staged(sc.n_planes + sc.n_spheres)), key
        assert np.array_equal(basis, h.camera_basis()), key


def test_staging_matches_golden(golden, scenes):
    z = golden.kat_basic
    for key, sc in scenes.items():
        inv, basis = stage_scene(sc)
        assert np.array_equal(inv, z[key + "/staged_strict"])
        assert np.array_equal(basis, z[key + "/basis_strict"])


def test_argument_validation(scenes):
    sc = scenes["spheres_a1"]
    for kw in (dict(width=0, height=4), dict(width=4, height=-1), dict(spp=0), dict(spp=70000), dict(depth_limit=-1),
               dict(tile=(12, 16)), dict(tile=(8, 8)), dict(tile_index=2, tile_count=2), dict(tile_count=-1)):
        args = dict(width=16, height=16)
        args.update(kw)
        with pytest.raises(capi.KajoError) as e:
            HipRenderer(sc, **args)
        assert e.value.code == -1, (kw, str(e.value))


def test_default_params_select_the_build_that_meets_the_north_star():
    """kajo_hip_default_params: the reference's constants (Renderer.cpp:21, Shader.cpp:24, Random.h:43) and EXACT numerics -- the fastest
    build whose frame stays within BASELINE's per-pixel RMSE < 1e-4 of the reference; FAST is an explicit choice (flags = KAJO_FLAG_FAST)."""
    p = capi.KajoParams()
    capi.lib().kajo_hip_default_params(C.byref(p))
    assert (p.samplesPerPass, p.depthLimit, p.seed, p.tileW, p.tileH, p.tileCount) == (32, 8, 0o715517, 64, 16, 1)
    assert p.flags == capi.KAJO_FLAG_EXACT and capi.KAJO_FLAG_FAST == 0
    header = open(os.path.join(ROOT, "include", "kajo_hip.h")).read()
    assert re.search(r"#define KAJO_EXACT_REL_TOL 1\.5e-3f", header) and re.search(r"#define KAJO_EXACT_ABS_FLOOR 1e-3f", header)


def test_scene_outside_the_exact_range_of_the_hand_made_division_is_refused(scenes):
    """STRICT / EXACT form the IEEE quotient and square root of the closest-hit walk without the compiler's range scaling
    (integrator.inc.hip kdiv / ksqrt): exact for scenes whose non-zero coordinates lie in 2^-40 .. 2^40, which kajo_hip_create checks --
    before it looks for a device, so the refusal is testable here. The reference divides in hardware (Raytracer.cpp:30-44) and has no
    such limit; FAST has none either."""
    import copy
    sc = scenes["spheres_a169"]
    for scale, bad in ((2.0 ** 50, True), (2.0 ** -50, True), (2.0 ** 30, False), (float("nan"), True)):
        moved = copy.deepcopy(sc)
        moved.spheres = sc.spheres.copy()
        moved.spheres[0, 12] = scale  # x of the first sphere's translation (column-major transform)
        for kw in (dict(strict=True), dict(exact=True)):
            with pytest.raises(capi.KajoError) as e:
                HipRenderer(moved, 16, 16, **kw)
            # (no GPU here: an accepted scene goes on to fail on the missing device)
            import torch
            if bad:
                assert e.value.code == capi.KAJO_E_INVALID and "2^-40 .. 2^40" in str(e.value), (scale, kw, str(e.value))
            elif not torch.cuda.is_available():
                assert e.value.code == capi.KAJO_E_NO_DEVICE, (scale, kw, str(e.value))
        if not torch.cuda.is_available():
            with pytest.raises(capi.KajoError) as e:
                HipRenderer(moved, 16, 16)  # FAST: no restriction
            assert e.value.code == capi.KAJO_E_NO_DEVICE


def test_no_cpu_fallback(scenes):
    """Without a GPU the product must fail loudly, not render on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(capi.KajoError) as e:
        HipRenderer(scenes["spheres_a1"], 16, 16)
    assert e.value.code == -3
    assert "no CPU path" in str(e.value)


def test_product_never_imports_oracle():
    """Nothing under kajo_amd/ may reference oracle/ (the oracle is test infrastructure)."""
    bad = []
    for dirpath, _, files in os.walk(os.path.join(ROOT, "kajo_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".h", ".hip", ".hpp")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                if re.search(r"oracle/|koracle_|kref_|libkajo_oracle|libkajo_ref", text):
                    bad.append(os.path.join(dirpath, f))
    assert not bad, bad


def test_product_library_has_no_environment_knobs():
    """Launch-shaping constants (hold thresholds, steal window, waves per block, split selection, grid density) are compile-time
    constants of libkajo_hip.so, as the reference's own are (renderer/cpu/Shader.cpp:23-24, Renderer.cpp:21): the library
    neither imports getenv nor contains a knob name. The tools' twin libkajo_hip_tune.so (kajo_amd/csrc/tuning.h) does."""
    import re
    import subprocess
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(here, "kajo_amd", "libkajo_hip.so")
    blob = open(lib, "rb").read()
    names = set(re.findall(rb"KAJO_[A-Z_]{3,}", blob)) - {b"KAJO_FLAG_COOP", b"KAJO_FLAG_DEFERRED"}  # (two error messages name the refused flags)
    assert not names, names
    syms = subprocess.run(["nm", "-D", "--undefined-only", lib], capture_output=True, text=True).stdout
    assert "getenv" not in syms
    tune = os.path.join(here, "kajo_amd", "libkajo_hip_tune.so")
    if os.path.exists(tune):
        assert b"KAJO_THR_L" in open(tune, "rb").read()
