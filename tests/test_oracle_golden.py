"""The CPU oracle reproduces the committed golden vectors (guards the oracle against drift)."""
import os

import numpy as np
import pytest

import common
import oracle_lib as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")
CASES = {
    "c2_hydrus6_se2": lambda: common.config2(n=8),
    "c2_hydrus6_se3": lambda: common.config2(n=8, orient="so3"),
    "c3_dragon4": lambda: common.config3(n=8),
    "c5_hydrus8": lambda: common.config5(n=8, n_boxes=200),
    "c1_hydrus4": lambda: common.config2(n=8, n_links=4, n_boxes=30),
}


@pytest.mark.parametrize("name", sorted(CASES))
def test_oracle_matches_golden(name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    w = CASES[name]()   # model / params only; states and boxes come from the fixture
    ref = O.eval_batch(w.model, g["boxes"], w.params, g["soa"], w.layout, threads=8)
    for k in ("flags", "band", "stable", "collision"):
        assert np.array_equal(ref[k], g[k]), k
    for k in ("min_dist", "margin", "fc_t_min", "fc_rp_min", "rp_position_margin", "cog", "gimbal"):
        assert np.allclose(ref[k], g[k], rtol=1e-12, atol=1e-13, equal_nan=True), k
    well = g["pivot_ratio"] > 1e-9
    assert np.allclose(ref["thrust"][:, well], g["thrust"][:, well], rtol=1e-9, atol=1e-10)


def test_oracle_edges_match_golden():
    g = np.load(os.path.join(GOLD, "c4_edges.npz"))
    model, params, _, _, _, _ = common.config4(n_edges=4)
    ok, first, band = O.check_motion_batch(model, g["boxes"], params, g["s1"], g["s2"], int(g["nd"]))
    assert np.array_equal(ok, g["ok"]) and np.array_equal(first, g["first"])
    ok, first, band = O.check_motion_batch(model, g["boxes"], params, g["s1"], g["s2"], g["nd_var"])
    assert np.array_equal(ok, g["ok_var"]) and np.array_equal(first, g["first_var"])
    sp = O.Space(params, model.n_joint)
    nd = [max(1, sp.valid_segment_count(g["s1"][:, e], g["s2"][:, e])) for e in range(64)]
    assert np.array_equal(nd, g["nd_var"][:64])
