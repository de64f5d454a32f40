"""Parity on the REAL workloads (BASELINE.json configs at their own frame size and pass count), not on thumbnails:

  configs[1]  data/spheres.json, 1920 x 1080, 16 passes x S = 32, depth 8
  configs[2]  data/spheres.json, 3840 x 2160, 64 passes x S = 32 (the multi-GPU frame, here on one GPU and as two tile owners)
  configs[3]  the caustics scene (ideal reflector + 3 lights), 1920 x 1080, 16 of its passes, depth 8
  configs[4]  the 1000-sphere / 16-light scene at its own 3840 x 2160, 2 passes (the oracle walks all 1000 spheres)
  configs[0]  256 x 256, S = 16, depth 1, one pass -- against the reference-produced frame of tests/golden/frames2.npz

The whole frame is rendered on the GPU through the C ABI; >= 8 crops of 64 x 32 pixels (glass sphere and its
silhouette, Phong / diffuse spheres, the emitter, the floor in shadow, the mirror wall, image corners) are rendered by
the oracle at the same passes (renderer/cpu/Renderer.cpp:36-75 under the stream protocol) and compared:
STRICT kernels bit for bit, FAST kernels within the stated tolerance with the number of pixels off by more than 1e-3
asserted. Every config is ALSO compared with crops rendered by the COMPILED REFERENCE itself (renderer/cpu/Renderer.cpp:36-75 on
the reference's own objects, both builds): configs[0], [1] in tests/golden/frames2.npz, configs[2], [3], [4] -- at 16 / 2 passes
and at their full 64 / 128 / 32 passes -- in tests/golden/frames3.npz (tests/golden/make_golden_frames3.py). STRICT must be
within clamped RMSE 1e-6 of the reference's -O2 build with a stated share of pixels bit-identical. What FAST measures against
the oracle at the full pass counts is written to gpurun_out/r06_parity_workloads.json (copied to profiles/r06_parity_workloads.json)."""
import json
import os
import zlib

import numpy as np
import pytest

from kajo_amd.renderer import HipRenderer
from kajo_amd.scene import stress_scene
from oraclelib import OracleLib, available
from workload_crops import crops_for

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not available("oracle"), reason="oracle not built")]
SEED = 0o715517
THREADS = max(1, min(16, os.cpu_count() or 1))


def scene_crc(sc):
    return zlib.crc32(np.ascontiguousarray(sc.planes).tobytes(), zlib.crc32(np.ascontiguousarray(sc.spheres).tobytes()))


def record(entry):
    """Append one measurement to gpurun_out/r06_parity_workloads.json (evidence; never read back by a test)."""
    try:
        d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "r06_parity_workloads.json")
        rows = json.load(open(path)) if os.path.exists(path) else []
        rows = [r for r in rows if r.get("key") != entry.get("key")] + [entry]
        json.dump(rows, open(path, "w"), indent=1)
    except OSError:
        pass


def clamped_rmse(a, b):
    m = np.isfinite(a) & np.isfinite(b)
    return float(np.sqrt(np.mean(np.where(m, np.clip(a, 0, 1) - np.clip(b, 0, 1), 0.0) ** 2)))


def check_against_reference_fixture(golden, key, scene, crops, passes, strict, fast, min_identical, fast_slack=1.5, exact=None):
    """The frames against crops the COMPILED REFERENCE rendered (tests/golden/frames3.npz): STRICT within clamped RMSE 1e-5 of
    the reference's -O2 build on every crop (BASELINE.json asks for 1e-4; measured <= 7.5e-6, profiles/r04_parity.json: the
    kernels associate the throughput product differently -- last-place differences -- and 1.5 % of their sin / cos values are the
    neighbour of glibc's, which flips a decision on about one path per crop of the many-light scenes) with at least
    `min_identical` of the pixels the reference's bit for bit; FAST within max(SURVEY tolerance, fast_slack x the difference of
    the reference's own two builds)."""
    z = golden.frames3
    assert int(z[key + "/scene_crc"]) == scene_crc(scene), "the scene generator changed: regenerate tests/golden/frames3.npz"
    assert int(z[key + "/passes"]) == passes
    rects = [tuple(int(v) for v in c[1:]) for c in crops]
    want_rects = [tuple(int(v) for v in r) for r in z[key + "/crops"]]
    assert rects[:len(want_rects)] == want_rects, (rects, want_rects)
    rows = []
    for k, (x, y, w, h) in enumerate(want_rects):
        ref_s, ref_f = z[key + "/rgb_crops_strict"][k], z[key + "/rgb_crops_fast"][k]
        floor = clamped_rmse(ref_s / passes, ref_f / passes)
        mm = np.isfinite(ref_s) & np.isfinite(ref_f)
        floor_med = float(np.median(np.abs(ref_s - ref_f)[mm] / passes))
        row = {"crop": json.loads(str(z[key + "/crop_names"]))[k], "reference_fastmath_vs_O2_rmse": floor, "reference_fastmath_vs_O2_median_abs": floor_med}
        if strict is not None:
            g = strict[y:y + h, x:x + w, :3]
            rm = clamped_rmse(g / passes, ref_s / passes)
            ident = float(np.mean(((g.view(np.uint32) == ref_s.view(np.uint32)) | (np.isnan(g) & np.isnan(ref_s))).all(-1)))
            row.update(strict_vs_reference_O2_rmse=rm, strict_px_bit_identical_to_reference_O2=ident)
            assert rm < 1e-5, (key, k, rm)
            assert ident >= min_identical, (key, k, ident)
        if exact is not None:
            # the EXACT build (what bench.py times) against the COMPILED REFERENCE's own crop: BASELINE.json's per-pixel RMSE < 1e-4,
            # asserted with a factor of five in hand (it decides as the oracle does; the oracle's strict math is within one decision per
            # crop of the reference's libm on the many-light scenes, as for STRICT above)
            g = exact[y:y + h, x:x + w, :3]
            rm = clamped_rmse(g / passes, ref_s / passes)
            row.update(exact_vs_reference_O2_rmse=rm)
            assert rm < 2e-5, (key, k, rm)
            assert np.array_equal(np.isfinite(g).all(-1), np.isfinite(ref_s).all(-1)), (key, k)
        if fast is not None:
            g = fast[y:y + h, x:x + w, :3]
            rm = clamped_rmse(g / passes, ref_s / passes)
            m = np.isfinite(g) & np.isfinite(ref_s)
            row.update(fast_vs_reference_O2_rmse=rm, fast_median_abs=float(np.median((np.abs(g - ref_s) / np.maximum(passes, np.abs(ref_s)))[m])))
            assert rm <= max(1e-3, fast_slack * floor), (key, k, rm, floor)
            # (the 1000-sphere scene: small spheres far from the origin, whose hit points binary32 resolves to ~1e-4 whatever the
            # formula -- the reference's own two builds differ by more than FAST differs from either)
            assert row["fast_median_abs"] <= max(1e-5, fast_slack * floor_med), (key, k, row)
        rows.append(row)
    record({"key": key + " vs compiled reference", "passes": passes, "crops": rows})
    return rows


def check_workload(scene, W, H, S, passes, depth, limit=10, fast_px_budget=0.004, ppl=0, slack=1.5, golden=None, fixture=None, min_identical=0.3, fast_north_star=None):
    crops = crops_for(scene, W, H, limit)
    assert len(crops) >= 8
    O = OracleLib("oracle")
    with HipRenderer(scene, W, H, spp=S, depth_limit=depth, seed=SEED, strict=True, passes_per_launch=ppl) as r:
        strict = r.render(passes).radiance()
    with HipRenderer(scene, W, H, spp=S, depth_limit=depth, seed=SEED, passes_per_launch=ppl) as r:
        fast = r.render(passes).radiance()
    with HipRenderer(scene, W, H, spp=S, depth_limit=depth, seed=SEED, exact=True, passes_per_launch=ppl) as r:
        exact = r.render(passes).radiance()
    hs, hf = O.create(scene, 1), O.create(scene, 0)
    # where the compiled reference travelled with the snapshot, its two builds (-O2 / fast-math) render every crop too:
    # what they differ by is the floor no implementation with other roundings can get under (an emitter's silhouette:
    # every path that flips there moves its pixel by emission / (25 * passes))
    refs = available("ref") and available("ref_strict")
    floors = {}
    if refs:  # the reference renders one rectangle per thread (its harness is single-threaded; ctypes releases the GIL)
        from concurrent.futures import ThreadPoolExecutor
        libs = {k: OracleLib(k) for k in ("ref", "ref_strict")}

        def ref_crop(job):
            k, (name, x, y, w, h) = job
            q = libs[k].create(scene)
            a = q.render(W, H, S=S, passes=passes, seed=SEED, depth_limit=depth, rect=(x, y, w, h))[y:y + h, x:x + w, :3] / passes
            q.close()
            return (k, name), a

        with ThreadPoolExecutor(THREADS) as ex:
            floors = dict(ex.map(ref_crop, [(k, c) for c in crops for k in ("ref", "ref_strict")]))
    report = []
    for name, x, y, w, h in crops:
        rect = (x, y, w, h)
        ws = hs.render(W, H, S=S, passes=passes, seed=SEED, depth_limit=depth, rect=rect, threads=THREADS)[y:y + h, x:x + w, :3]
        gs = strict[y:y + h, x:x + w, :3]
        same = (gs.view(np.uint32) == ws.view(np.uint32)) | (np.isnan(gs) & np.isnan(ws))
        assert same.all(), "%s %s: STRICT differs from the oracle in %d channels" % (scene.name, name, (~same).sum())
        ge = exact[y:y + h, x:x + w, :3]  # EXACT: the same pixels not-a-number, the rest within rounding of the oracle's
        assert np.array_equal(np.isfinite(ge).all(-1), np.isfinite(ws).all(-1)), (scene.name, name)
        assert clamped_rmse(ge / passes, ws / passes) < 1e-5, (scene.name, name, clamped_rmse(ge / passes, ws / passes))
        wf = hf.render(W, H, S=S, passes=passes, seed=SEED, depth_limit=depth, rect=rect, threads=THREADS)[y:y + h, x:x + w, :3] / passes
        gf = fast[y:y + h, x:x + w, :3] / passes
        m = np.isfinite(gf) & np.isfinite(wf)
        # (relative where the radiance is above 1 -- the crops on the lights: FAST adds the passes in groups of four,
        # tests/test_hip_tail_parts.py, and one unit in the last place of a total of 300 is 3e-5)
        d = (np.abs(gf - wf) / np.maximum(1.0, np.abs(wf)))[m]
        cl = np.where(m, np.clip(gf, 0, 1) - np.clip(wf, 0, 1), 0.0)
        off = int((np.abs(cl).max(-1) > 1e-3).sum())
        floor = (0.0, 0.0, 0.0, 0)
        if refs:
            ra, rb = floors[("ref", name)], floors[("ref_strict", name)]
            mm = np.isfinite(ra) & np.isfinite(rb)
            dd = np.abs(ra - rb)[mm]
            cc = np.where(mm, np.clip(ra, 0, 1) - np.clip(rb, 0, 1), 0.0)
            floor = (float(np.median(dd)), float(np.percentile(dd, 99)), float(np.sqrt(np.mean(cc ** 2))), int((np.abs(cc).max(-1) > 1e-3).sum()))
        # NaN pixels are the reference's own (inf * 0 at exactly-grazing glass hits): counted on both sides
        nonfinite = (int((~np.isfinite(gf)).any(-1).sum()), int((~np.isfinite(wf)).any(-1).sum()))
        report.append((name, float(np.median(d)), float(np.percentile(d, 99)), float(np.sqrt(np.mean(cl ** 2))), off, nonfinite, floor))
    # FAST: SURVEY section 8c tolerances on every crop (median 1e-5, p99 2e-3, clamped RMSE 1e-3) ...
    # (or 1.5 x the reference's own two-build difference on that crop, whichever is larger)
    slack = slack if refs else 2.5
    for name, med, p99, rmse, off, nonfinite, floor in report:
        tol = [max(t, slack * f) if refs else slack * t for t, f in zip((1e-5, 2e-3, 1e-3), floor)]
        assert med <= tol[0] and p99 <= tol[1] and rmse <= tol[2], (scene.name, name, (med, p99, rmse), floor)
        assert abs(nonfinite[0] - nonfinite[1]) <= 2 and nonfinite[0] <= 2 + 2 * nonfinite[1], (scene.name, name, nonfinite)
    # ... and the pixels that part from the oracle by more than 1e-3 (a decision flipped at an ill-conditioned hit:
    # DESIGN.md section 2) stay a handful: at most `fast_px_budget` of the compared pixels
    total_off, floor_off = sum(r[4] for r in report), sum(r[6][3] for r in report)
    assert total_off <= max(fast_px_budget * len(crops) * 64 * 32, slack * floor_off), (scene.name, total_off, floor_off, report)
    print("%s %dx%d x%d: FAST px off by > 1e-3: %d of %d (reference -O2 vs fast-math: %d)" % (scene.name, W, H, passes, total_off, len(crops) * 64 * 32, floor_off))
    cl_all = np.sqrt(np.mean([r[3] ** 2 for r in report]))
    record({"key": "%s %dx%d FAST vs oracle(libm) at %d passes" % (scene.name, W, H, passes), "frame": "%dx%d" % (W, H), "passes": passes,
            "fast_rmse_over_crops": float(cl_all), "meets_1e-4": bool(cl_all < 1e-4),
            "crops": [{"crop": r[0], "fast_vs_oracle_libm_rmse": r[3], "px_off_by_more_than_1e-3": r[4], "reference_fastmath_vs_O2_rmse": r[6][2]} for r in report]})
    if fast_north_star is not None:  # BASELINE.json's per-pixel RMSE bound, where the pass count lets FAST meet it (profiles/r04_parity.json)
        assert cl_all < fast_north_star, (scene.name, passes, float(cl_all))
    if refs:  # the reference libraries travelled: STRICT against the reference's -O2 build rendered here, crop by crop
        for name, x, y, w, h in crops:
            g, ref = strict[y:y + h, x:x + w, :3], floors[("ref_strict", name)] * passes
            assert clamped_rmse(g / passes, ref / passes) < 1e-5, (scene.name, name)
    if fixture:  # ... and against the committed crops of the compiled reference, wherever this runs
        check_against_reference_fixture(golden, fixture, scene, crops, passes, strict, fast, min_identical, fast_slack=slack, exact=exact)
    return report


def test_configs1_spheres_1080p_16_passes(scenes, golden):
    sc = scenes["spheres_a169"]
    check_workload(sc, 1920, 1080, 32, 16, 8)
    # the same frame against crops rendered by the COMPILED REFERENCE (-O2 build and fast-math build)
    z = golden.frames2
    crops = z["c2_1080p/crops"]
    with HipRenderer(sc, 1920, 1080, spp=32, depth_limit=8, seed=int(z["seed"]), strict=True) as r:
        strict = r.render(16).radiance()
    with HipRenderer(sc, 1920, 1080, spp=32, depth_limit=8, seed=int(z["seed"])) as r:
        fast = r.render(16).radiance()
    with HipRenderer(sc, 1920, 1080, spp=32, depth_limit=8, seed=int(z["seed"]), exact=True) as r:
        exact = r.render(16).radiance()
    for k, (x, y, w, h) in enumerate(crops):
        ref_s, ref_f = z["c2_1080p/rgb_crops_strict"][k] / 16, z["c2_1080p/rgb_crops_fast"][k] / 16
        floor = np.sqrt(np.nanmean((np.clip(ref_s, 0, 1) - np.clip(ref_f, 0, 1)) ** 2))  # the reference against itself
        for got, scale in ((strict, 1.0), (exact, 1.0), (fast, 1.5)):
            g = got[y:y + h, x:x + w, :3] / 16
            m = np.isfinite(g) & np.isfinite(ref_s)
            rmse = np.sqrt(np.mean(((np.clip(g, 0, 1) - np.clip(ref_s, 0, 1)) ** 2)[m]))
            assert np.median((np.abs(g - ref_s) / np.maximum(1.0, np.abs(ref_s)))[m]) <= 1e-5  # (relative above 1: the crops on the lights)
            assert rmse <= max(1e-3, scale * floor), (k, rmse, floor)
            if got is strict:
                # BASELINE.json's figure, against the COMPILED REFERENCE (its -O2 build) on its own frame: per-pixel RMSE < 1e-4.
                # Asserted with two decades of margin (measured 2.9e-8, profiles/r02_parity.json: the kernels differ from that
                # build only in the association of the throughput product and in 1.5 % of the sin/cos values, by one ulp)
                assert rmse < 1e-6, (k, rmse)
            if got is exact:
                # ... and the EXACT build (the one bench.py times) against the compiled reference on the same crops: two decades inside 1e-4
                assert rmse < 2e-6, (k, rmse)
        # STRICT evaluates the reference's -O2 arithmetic: most pixels of a crop are the reference's, bit for bit
        g = strict[y:y + h, x:x + w, :3]
        assert np.mean((g.view(np.uint32) == z["c2_1080p/rgb_crops_strict"][k].view(np.uint32)).all(-1)) >= 0.4


def test_configs2_spheres_4k_64_passes(scenes, golden):
    """The multi-GPU frame at its own size and pass count, all 64 passes in one launch as bench.py renders it."""
    sc = scenes["spheres_a169"]
    # at configs[2]'s 64 passes FAST meets the north-star bound itself: 4.1e-5 over the eight crops (4.7e-5 on a whole small frame)
    check_workload(sc, 3840, 2160, 32, 64, 8, limit=8, ppl=64, golden=golden, fixture="c3_4k", fast_north_star=1e-4)
    # and as rank 0 of two tile owners would render it: the owned half of the tiles, bit for bit the one-owner frame's
    crops = crops_for(sc, 3840, 2160, 8)
    with HipRenderer(sc, 3840, 2160, spp=32, depth_limit=8, seed=SEED, passes_per_launch=64) as r:
        whole = r.render(64).radiance()
    import ctypes as C
    from kajo_amd import capi
    from kajo_amd.tiles import TileLayout
    lay = TileLayout(3840, 2160, 2)
    bufs = []
    for rank in range(2):
        with HipRenderer(sc, 3840, 2160, spp=32, depth_limit=8, seed=SEED, passes_per_launch=64, tile_index=rank, tile_count=2) as r:
            r.render(64).wait()
            ptr, nbytes = r.tile_buffer()
            host = np.empty(nbytes // 4, np.float32)
            import torch
            t = torch.as_tensor(host)
            capi_lib = capi.lib()
            # device -> host copy of the compact tile buffer through torch (plumbing)
            from bench import DevicePtr
            t.copy_(torch.as_tensor(DevicePtr(ptr, nbytes // 4), device="cuda"))
            bufs.append(host.reshape(-1, 4))
    frame = lay.compose(np.stack(bufs))
    for name, x, y, w, h in crops:
        assert np.array_equal(frame[y:y + h, x:x + w].view(np.uint32), whole[y:y + h, x:x + w].view(np.uint32)), name


def test_configs3_caustics_1080p(scenes, golden):
    check_workload(scenes["caustics_a169"], 1920, 1080, 32, 16, 8, fast_px_budget=0.008, golden=golden, fixture="c4_1080p")


def test_configs4_stress_1000_spheres_4k(scenes, golden):
    sc = stress_scene(scenes["spheres_a169"], 1000, 16)
    # two passes: the oracle and the reference walk all 1006 primitives per ray (CPU minutes at more). 50 paths per pixel
    # among 1000 small Phong / diffuse spheres and 16 lights leave single flipped paths visible: measured 1.2-1.8 x the
    # reference's own two-build difference (as in test_hip_edge_cases.py), hence the wider slack
    check_workload(sc, 3840, 2160, 32, 2, 8, limit=8, fast_px_budget=0.01, ppl=2, slack=2.5, golden=golden, fixture="c5_4k")


def check_full_pass_count(golden, fixture, scene, W, H, S, passes, depth, ncrops, ppl, min_identical=0.3, fast_north_star=None):
    """A workload at its FULL pass count on `ncrops` feature crops: STRICT = oracle bit for bit, STRICT and FAST against the
    crops the compiled reference rendered at that pass count (frames3.npz), and FAST's clamped RMSE against the oracle recorded
    (a flipped path moves its pixel by 1 / (25 passes), so the figure falls with the pass count). `fast_north_star`: assert
    that FAST's RMSE over the crops is below it (BASELINE.json: 1e-4) where the measurement supports that."""
    crops = crops_for(scene, W, H, 10)[:ncrops]
    with HipRenderer(scene, W, H, spp=S, depth_limit=depth, seed=SEED, strict=True, passes_per_launch=ppl) as r:
        strict = r.render(passes).radiance()
    with HipRenderer(scene, W, H, spp=S, depth_limit=depth, seed=SEED, passes_per_launch=ppl) as r:
        fast = r.render(passes).radiance()
    with HipRenderer(scene, W, H, spp=S, depth_limit=depth, seed=SEED, exact=True, passes_per_launch=ppl) as r:
        exact = r.render(passes).radiance()
    O = OracleLib("oracle")
    hs, hf = O.create(scene, 1), O.create(scene, 0)
    sq, n, rows = 0.0, 0, []
    for name, x, y, w, h in crops:
        ws = hs.render(W, H, S=S, passes=passes, seed=SEED, depth_limit=depth, rect=(x, y, w, h), threads=THREADS)[y:y + h, x:x + w, :3]
        gs = strict[y:y + h, x:x + w, :3]
        same = (gs.view(np.uint32) == ws.view(np.uint32)) | (np.isnan(gs) & np.isnan(ws))
        assert same.all(), "%s %s at %d passes: STRICT differs from the oracle in %d channels" % (scene.name, name, passes, (~same).sum())
        wf = hf.render(W, H, S=S, passes=passes, seed=SEED, depth_limit=depth, rect=(x, y, w, h), threads=THREADS)[y:y + h, x:x + w, :3] / passes
        gf = fast[y:y + h, x:x + w, :3] / passes
        m = np.isfinite(gf) & np.isfinite(wf)
        cl = np.where(m, np.clip(gf, 0, 1) - np.clip(wf, 0, 1), 0.0)
        sq += float((cl ** 2).sum())
        n += cl.size
        rows.append({"crop": name, "fast_vs_oracle_libm_rmse": float(np.sqrt(np.mean(cl ** 2))), "px_off_by_more_than_1e-3": int((np.abs(cl).max(-1) > 1e-3).sum())})
    rmse = float(np.sqrt(sq / n))
    record({"key": "%s FAST vs oracle(libm) at %d passes" % (fixture, passes), "frame": "%dx%d" % (W, H), "passes": passes,
            "fast_rmse_over_crops": rmse, "meets_1e-4": bool(rmse < 1e-4), "crops": rows})
    print("%s %dx%d x %d passes: FAST vs oracle clamped RMSE over %d feature crops %.3g" % (scene.name, W, H, passes, len(crops), rmse))
    if fast_north_star is not None:
        assert rmse < fast_north_star, (scene.name, passes, rmse)
    check_against_reference_fixture(golden, fixture, scene, crops, passes, strict, fast, min_identical, exact=exact)


def test_configs3_caustics_at_all_128_passes(scenes, golden):
    """configs[3] at its own 4096 spp = 128 passes x S = 32 (Renderer.cpp:44-72 runs the pass loop that long): three crops."""
    # (128 passes of 25 paths: 3200 last-place differences to collect per pixel -- a quarter of the pixels still are the reference's bit for bit)
    # (FAST over these crops at 128 passes: 4.2e-5)
    check_full_pass_count(golden, "c4_1080p_128", scenes["caustics_a169"], 1920, 1080, 32, 128, 8, 3, 16, min_identical=0.15, fast_north_star=1e-4)


def test_configs4_stress_at_all_32_passes(scenes, golden):
    """configs[4] at its own 1024 spp = 32 passes x S = 32, all in one launch as tools/configs.py times it: two crops."""
    check_full_pass_count(golden, "c5_4k_32", stress_scene(scenes["spheres_a169"], 1000, 16), 3840, 2160, 32, 32, 8, 2, 32, min_identical=0.15)


def test_configs0_c1_full_size_against_the_reference(scenes, golden):
    """BASELINE configs[0] exactly (256 x 256, 16 spp, 1 bounce): ARGB8 of the whole frame and float crops produced by the
    compiled reference; plus the 64-pass converged frame."""
    z = golden.frames2
    sc = scenes["spheres_a1"]
    seed = int(z["seed"])
    for strict in (True, False):
        with HipRenderer(sc, 256, 256, spp=16, depth_limit=1, seed=seed, strict=strict) as r:
            acc = r.render(1).radiance()
            argb = r.argb8()
        for tag in ("strict", "fast"):
            want = z["c1_256/argb8_" + tag]
            ch = lambda a, s: ((a >> s) & 255).astype(np.int32)
            dmax = np.maximum.reduce([np.abs(ch(argb, s) - ch(want, s)) for s in (0, 8, 16)])
            assert np.mean(dmax > 1) <= 0.002, (strict, tag, float(np.mean(dmax > 1)))
        for k, (x, y, w, h) in enumerate(z["c1_256/crops"]):
            ref = z["c1_256/rgb_crops_strict"][k]
            g = acc[y:y + h, x:x + w, :3]
            m = np.isfinite(g) & np.isfinite(ref)
            reff = z["c1_256/rgb_crops_fast"][k]
            floor99 = np.percentile(np.abs(ref - reff)[np.isfinite(ref) & np.isfinite(reff)], 99)  # the reference's two builds
            # (the median relative to the value where that exceeds 1: the emitter's pixels hold 445, whose last place is 3e-5)
            assert np.median((np.abs(g - ref) / np.maximum(1.0, np.abs(ref)))[m]) <= 1e-5
            assert np.percentile(np.abs(g - ref)[m], 99) <= max(2e-3, 1.5 * floor99)
            if strict:
                assert np.mean((g.view(np.uint32) == ref.view(np.uint32)).all(-1)) >= 0.9
                # the north-star figure against the compiled reference, with two decades of margin (as for configs[1])
                assert np.sqrt(np.mean(((np.clip(g, 0, 1) - np.clip(ref, 0, 1)) ** 2)[m])) < 1e-6
        with HipRenderer(sc, 64, 64, spp=32, depth_limit=8, seed=seed, strict=strict) as r:
            conv = r.render(64).radiance()[..., :3] / 64
        ref_s, ref_f = z["conv_64/rgb_strict"] / 64, z["conv_64/rgb_fast"] / 64
        m = np.isfinite(conv) & np.isfinite(ref_s)
        rmse = np.sqrt(np.mean(((np.clip(conv, 0, 1) - np.clip(ref_s, 0, 1)) ** 2)[m]))
        floor = np.sqrt(np.nanmean((np.clip(ref_s, 0, 1) - np.clip(ref_f, 0, 1)) ** 2))
        assert rmse <= max(1e-3, 1.5 * floor), (strict, rmse, floor)


def test_configs1_whole_frame_strict_equals_oracle(scenes):
    """BASELINE configs[1] exactly as bench.py times it -- 1920 x 1080, 16 passes x S = 32, depth 8, one launch -- and the WHOLE frame:
    every one of the 2 073 600 pixels of the STRICT kernels' accumulation buffer equals the oracle's, bit for bit (the oracle takes
    the host cores ~20 s for the 829 M paths); FAST over the whole frame is recorded beside it."""
    import time
    sc = scenes["spheres_a169"]
    W, H, P = 1920, 1080, 16
    t0 = time.time()
    want = OracleLib("oracle").create(sc, 1).render(W, H, S=32, passes=P, seed=SEED, depth_limit=8, threads=THREADS)
    t_oracle = time.time() - t0
    with HipRenderer(sc, W, H, spp=32, depth_limit=8, seed=SEED, strict=True, passes_per_launch=16) as r:
        got = r.render(P).radiance()
    same = ((got.view(np.uint32) == want.view(np.uint32)) | (np.isnan(got) & np.isnan(want)))[..., :3].all(-1)
    assert same.all(), "%d of %d pixels differ" % (int((~same).sum()), same.size)
    # EXACT (the build bench.py times): the same pixels not-a-number, every other within rounding of the oracle's -- no path of the
    # 829 M decided differently (a flipped path moves its pixel by a path's radiance / 400: FAST has ~900 such pixels)
    with HipRenderer(sc, W, H, spp=32, depth_limit=8, seed=SEED, exact=True, passes_per_launch=16) as r:
        ex = r.render(P).radiance()
    assert np.array_equal(~np.isfinite(ex[..., :3]).all(-1), ~np.isfinite(want[..., :3]).all(-1))
    ex_rmse = clamped_rmse(ex[..., :3] / P, want[..., :3] / P)
    ex_off = int((np.abs(np.clip(ex[..., :3] / P, 0, 1) - np.clip(want[..., :3] / P, 0, 1)).max(-1) > 1e-3).sum())
    assert ex_rmse < 1e-6 and ex_off == 0, (ex_rmse, ex_off)
    from exact_tol import assert_exact_within_tolerance
    tol = assert_exact_within_tolerance(ex, want, P, "configs[1]")  # include/kajo_hip.h: per channel within 1.5e-3 of max(|oracle|, 1e-3)
    record({"key": "configs[1] whole frame, EXACT", "exact_vs_oracle_strict_rmse_whole_frame": ex_rmse, "exact_px_off_by_more_than_1e-3": ex_off,
            "exact_vs_oracle_rmse_linear": tol["rmse_linear"], "exact_vs_oracle_max_rel": tol["max_rel"]})
    del ex
    wantf = OracleLib("oracle").create(sc, 0).render(W, H, S=32, passes=P, seed=SEED, depth_limit=8, threads=THREADS)[..., :3] / P
    with HipRenderer(sc, W, H, spp=32, depth_limit=8, seed=SEED, passes_per_launch=16) as r:
        fast = r.render(P).radiance()[..., :3] / P
    rm = clamped_rmse(fast, wantf)
    off = int((np.abs(np.clip(fast, 0, 1) - np.clip(wantf, 0, 1)).max(-1) > 1e-3).sum())
    record({"key": "configs[1] whole frame 1920x1080 x 16 passes", "strict_px_bit_identical_to_oracle": int(same.sum()), "px": int(same.size),
            "fast_vs_oracle_libm_rmse_whole_frame": rm, "fast_px_off_by_more_than_1e-3": off, "oracle_seconds": round(t_oracle, 1), "oracle_threads": THREADS})
    print("configs[1] whole frame: STRICT %d / %d px bit-identical; FAST clamped RMSE %.3g, %d px off by > 1e-3 (oracle %.1f s on %d threads)" % (
        int(same.sum()), same.size, rm, off, t_oracle, THREADS))
    assert rm < 1e-3
