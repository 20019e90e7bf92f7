"""GPU: batched B-spline evaluation (mvk_spline_eval_dev) against the host evaluation and scipy-pinned values.
Reference: BsplineRos::evaluate (bspline_ros/src/bspline_ros.cpp:88-110).  Tolerance 1e-12 absolute (fp64, same
de Boor recurrence; FMA contraction differs)."""
import numpy as np
import pytest

import motion_planning_amd as mp
from motion_planning_amd import path as PP
from test_path_post import _random_path

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("degree,J,nt", [(5, 5, 4001), (3, 3, 63), (5, 7, 1), (7, 6, 1000)])
def test_device_spline_evaluation_matches_host(degree, J, nt):
    import torch
    rng = np.random.default_rng(degree + J)
    path = _random_path(rng, 40, J)
    period = 100.0
    sp = PP.Spline(path, period, degree)
    t = np.linspace(-1.0, period + 1.0, nt) if nt > 1 else np.array([37.5])
    d_t = torch.from_numpy(t).cuda()
    for k in range(0, min(degree, 3)):
        got = sp.eval_dev(d_t, k).cpu().numpy().T
        ref = sp.eval(t, k)
        scale = max(1.0, np.abs(ref).max())
        assert np.allclose(got, ref, rtol=0, atol=1e-12 * scale), (k, np.abs(got - ref).max())
    with pytest.raises(mp.MvkError):
        sp.eval_dev(d_t, degree)
    sp.close()


@pytest.mark.parametrize("degree", [5, 3])
def test_device_spline_evaluation_matches_the_scipy_golden_vectors(degree):
    """tests/golden/spline_d*.npz: scipy.interpolate.BSpline values (made by tests/golden/make_spline_golden.py)"""
    import os

    import torch
    g = np.load(os.path.join(os.path.dirname(__file__), "golden", f"spline_d{degree}.npz"))
    sp = PP.Spline(g["path"], float(g["period"]), int(g["degree"]))
    d_t = torch.from_numpy(g["t"]).cuda()
    nc = len(g["path"]) + 2
    for k in range(3):
        got = sp.eval_dev(d_t, k).cpu().numpy().T
        ref = g[f"d{k}"]
        tol = 1e-10 if k == 0 else 1e-8 * max(1.0, (nc / float(g["period"])) ** k) + 1e-8 * np.abs(ref).max()
        assert np.allclose(got, ref, rtol=0, atol=tol), (k, np.abs(got - ref).max())
    sp.close()
