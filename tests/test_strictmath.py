"""include/kajo_strictmath.h: within 1 ulp of the correctly rounded result on the domains the integrator
uses (CPU, through the oracle library), and bit-identical on the GPU (gpu-marked, through the C ABI).
tools/strictmath_exhaustive.c makes the same check over EVERY binary32 of the domains (minutes of CPU time)."""
import ctypes as C

import numpy as np
import pytest

from oraclelib import OracleLib, available

pytestmark = pytest.mark.skipif(not available("oracle"), reason="oracle not built")

RNG = np.random.default_rng(7)
N = 200000
U = RNG.random(N).astype(np.float32)
CASES = [
    (0, (2 * np.pi * U).astype(np.float32), None, np.sin),                 # sin(2 pi s), Light.cpp:43, Random.cpp:84
    (0, (np.pi * (U - .5)).astype(np.float32), None, np.sin),              # sin(pi (s - .5)), Light.cpp:45
    (1, (2 * np.pi * U).astype(np.float32), None, np.cos),
    (2, (2 * U - 1).astype(np.float32), None, np.arcsin),                  # asin(r / dist), Light.cpp:32
    (3, (2 * U - 1).astype(np.float32), None, np.arccos),                  # acos(u^(1/(e+1))), Random.cpp:95
    (4, U, np.full(N, 100, np.float32), np.power),                         # cos^e, BSDF.cpp:66
    (4, U, np.full(N, 1 / 101, np.float32), np.power),                     # u^(1/(e+1))
    (4, U, np.full(N, 1 / 2.2, np.float32), np.power),                     # linearToSRGB, Image.cpp:16
    (4, (16 * U).astype(np.float32), np.full(N, 2.2, np.float32), np.power),  # srgbToLinear, Parser.cpp:72
]


def cpu_eval(fn, x, y):
    L = OracleLib("oracle").lib
    out = np.zeros_like(x)
    yy = x if y is None else y
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    L.koracle_strictmath(C.c_int(fn), C.c_int(x.size), p(x), p(yy), p(out))
    return out


def ulp_distance(a, b):
    ia = a.view(np.int32).astype(np.int64)
    ib = b.view(np.int32).astype(np.int64)
    ia = np.where(ia < 0, -(ia & 0x7fffffff), ia)
    ib = np.where(ib < 0, -(ib & 0x7fffffff), ib)
    return np.abs(ia - ib)


@pytest.mark.parametrize("case", range(len(CASES)))
def test_within_one_ulp_of_correct_rounding(case):
    fn, x, y, ref = CASES[case]
    got = cpu_eval(fn, x, y)
    want = (ref(x.astype(np.float64)) if y is None else ref(x.astype(np.float64), y.astype(np.float64))).astype(np.float32)
    ok = np.isfinite(want) & (np.abs(want) > 1e-37)  # denormal results: compare absolutely below
    assert ulp_distance(got[ok], want[ok]).max() <= 1
    assert np.abs(got[~ok] - want[~ok]).max(initial=0) <= 1e-37
    # and almost always exactly the correctly rounded value: the binary32 evaluation of sin / cos / asin / acos rounds
    # a value that is itself good to ~0.01 ulp (98.5 % of uniformly drawn arguments; 99.9 % of all binary32 arguments,
    # which crowd towards zero), the binary64 evaluation of pow one that is good to ~1e-4 ulp
    assert np.mean(got[ok] == want[ok]) > (0.999 if fn == 4 else 0.98)


def test_special_values():
    x = np.array([0, 0, 1, 0.5, np.nan, -1, 0, 4], np.float32)
    y = np.array([100, 0, 100, 0, 2, 2, -1, .5], np.float32)
    got = cpu_eval(4, x, y)
    assert got[0] == 0 and got[1] == 1 and got[2] == 1 and got[3] == 1
    assert np.isnan(got[4]) and np.isnan(got[5]) and np.isinf(got[6]) and got[7] == 2
    assert cpu_eval(3, np.array([1, -1], np.float32), None)[0] == 0
    assert np.isnan(cpu_eval(2, np.array([1.0000001], np.float32), None)[0])


@pytest.mark.gpu
@pytest.mark.parametrize("case", range(len(CASES)))
def test_gpu_bits_equal_cpu_bits(case, scenes):
    from kajo_amd import capi
    from kajo_amd.renderer import HipRenderer
    fn, x, y, _ = CASES[case]
    yy = x if y is None else y
    out = np.zeros_like(x)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    with HipRenderer(scenes["spheres_a1"], 8, 8, strict=True) as r:
        capi.check(capi.lib().kajo_hip_kat_strictmath(r._h, fn, x.size, p(x), p(yy), p(out)))
    assert np.array_equal(out.view(np.uint32), cpu_eval(fn, x, y).view(np.uint32))
