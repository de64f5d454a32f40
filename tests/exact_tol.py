"""EXACT's stated tolerance against the oracle (include/kajo_hip.h KAJO_EXACT_REL_TOL / KAJO_EXACT_ABS_FLOOR), as numbers the whole-frame
comparisons assert: the same pixels not-a-number; every other channel of the estimate within REL_TOL of max(|oracle|, ABS_FLOOR)."""
import os
import re

import numpy as np

_HEADER = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "kajo_hip.h")).read()
REL_TOL = float(re.search(r"#define KAJO_EXACT_REL_TOL ([0-9.e+-]+)f", _HEADER).group(1))
ABS_FLOOR = float(re.search(r"#define KAJO_EXACT_ABS_FLOOR ([0-9.e+-]+)f", _HEADER).group(1))


def exact_figures(got_sum, want_sum, passes):
    """(H, W, >= 3) sums over passes -> dict(max_rel, rmse_linear, rmse_clamped); asserts the NaN pixels agree."""
    g, w = got_sum[..., :3].astype(np.float64) / passes, want_sum[..., :3].astype(np.float64) / passes
    nan_g, nan_w = ~np.isfinite(g).all(-1), ~np.isfinite(w).all(-1)
    assert np.array_equal(nan_g, nan_w), "not-a-number pixels differ: %d here, %d in the oracle" % (nan_g.sum(), nan_w.sum())
    m = ~nan_w
    d = g[m] - w[m]
    return dict(max_rel=float((np.abs(d) / np.maximum(np.abs(w[m]), ABS_FLOOR)).max()), rmse_linear=float(np.sqrt(np.mean(d ** 2))),
                rmse_clamped=float(np.sqrt(np.mean((np.clip(g[m], 0, 1) - np.clip(w[m], 0, 1)) ** 2))), nan_px=int(nan_w.sum()))


def assert_exact_within_tolerance(got_sum, want_sum, passes, what=""):
    f = exact_figures(got_sum, want_sum, passes)
    assert f["max_rel"] <= REL_TOL, (what, f)
    return f
