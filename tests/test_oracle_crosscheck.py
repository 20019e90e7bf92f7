"""The C oracle against an independently written NumPy/SciPy restatement of V-spec (oracle/np_vspec.py).

Guards the oracle against its own typos; it does not pin parity with the absent third-party reference
packages (that part stays "parity unpinned", see the oracle's header).
"""
import os
import sys

import numpy as np
import pytest

import common
import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import np_vspec as NP  # noqa: E402


@pytest.mark.parametrize("cfg", ["c2", "c2se3", "c3", "c5"])
def test_model_quantities(cfg):
    n = 60
    w = {"c2": lambda: common.config2(n=n), "c2se3": lambda: common.config2(n=n, orient="so3"),
         "c3": lambda: common.config3(n=n, n_boxes=5), "c5": lambda: common.config5(n=n, n_boxes=5)}[cfg]()
    ref = O.eval_batch(w.model, w.boxes, w.params, w.soa, w.layout)
    for i in range(n):
        q = NP.model_update(w.model, w.soa[:, i])
        assert np.allclose(q["cog"], ref["cog"][:, i], rtol=1e-11, atol=1e-12)
        assert q["fc_t_min"] == pytest.approx(ref["fc_t_min"][i], rel=1e-9, abs=1e-10)
        assert q["fc_rp_min"] == pytest.approx(ref["fc_rp_min"][i], rel=1e-9, abs=1e-10)
        assert q["rp_position_margin"] == pytest.approx(ref["rp_position_margin"][i], rel=1e-8, abs=1e-9)
        if q["cond"] < 1e6:
            assert np.allclose(q["thrust"], ref["thrust"][:, i], rtol=1e-7, atol=1e-7)
            assert q["margin"] == pytest.approx(ref["margin"][i], rel=1e-7, abs=1e-7)
        if w.model.has_gimbal:
            assert np.allclose(q["gimbal"], ref["gimbal"][:, i], atol=1e-11)
        c, R = O.prim_poses(w.model, w.params, w.soa[:, i])
        for k, T in enumerate(q["prims"]):
            assert np.allclose(T[:3, 3], c[k], atol=1e-12) and np.allclose(T[:3, :3], R[k], atol=1e-12)


def test_narrowphase_against_generic_optimisers():
    rng = np.random.default_rng(5)
    checked = 0
    for _ in range(400):
        ptype = int(rng.integers(0, 3))
        q = rng.normal(size=4)
        q /= np.linalg.norm(q)
        T = NP.quat_to_T(q, rng.uniform(-1.5, 1.5, 3))
        dim = [rng.uniform(0.05, 0.3), rng.uniform(0.05, 0.4), rng.uniform(0.02, 0.3)]
        bq = rng.normal(size=4)
        bq /= np.linalg.norm(bq)
        box = np.r_[rng.uniform(-0.5, 0.5, 3), bq, rng.uniform(0.1, 1.2, 3)]
        d = O.prim_box_signed_dist(ptype, T[:3, 3], T[:3, :3], dim, box)
        d_opt = NP.prim_box_dist_opt(ptype, T, dim, box)
        if d > 1e-6:
            assert d == pytest.approx(d_opt, rel=2e-6, abs=2e-7), (ptype, d, d_opt)
            checked += 1
        else:
            assert d_opt < 2e-6, (ptype, d, d_opt)   # overlapping per the oracle => optimiser finds ~0
    assert checked > 100


def test_band_reporting():
    """a sample placed within 1e-6 m of contact is flagged as band"""
    m = mp.make_model("hydrus", 4)
    p = mp.default_params(m, A.MODE_ONLY_BASE, A.SE2)
    s = np.array([[0, 0, 0, 0, 0, 0, 1.0, 0, 0, 0]]).T   # straight chain along +x, link4 box ends at x = 2.4
    for gap, band, coll in ((5e-7, 1, 0), (-5e-7, 1, 1), (1e-3, 0, 0), (-1e-3, 0, 1)):
        box = np.array([[2.4 + gap + 0.5, 0, 0, 0, 0, 0, 1, 1.0, 0.02, 0.02]])
        r = O.eval_batch(m, box, p, s)
        assert (r["band"][0], r["collision"][0]) == (band, coll), (gap, r["min_dist"][0])
        assert r["min_dist"][0] == pytest.approx(gap, abs=1e-12)
