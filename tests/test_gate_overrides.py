"""Row a10 and the gate overrides of mvk_params.

The reference gates the two stages by (planning mode, motion type): stability iff mode in {0, 2} or SE(3)
(motion_planning.cpp:138-151), collision iff mode != 0 (:153).  mvk_params::check_stability / check_collision = -1
reproduce that; 0 / 1 force a stage off / on.  MotionPlanning::checkCollision(const MultilinkState&)
(motion_planning.cpp:802-810) -- collision of an arbitrary state, no stability, whatever the mode -- is the setting
(check_stability, check_collision) = (0, 1) read through mvk_result::collision (quirk Q8, "always returns true", is
not reproduced)."""
import numpy as np
import pytest

import common
import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A

COMBOS = [("hydrus", 0, 0), ("hydrus", 1, 0), ("hydrus", 1, 1), ("hydrus", 2, 0), ("hydrus", 2, 1), ("dragon", 1, 1), ("dragon", 2, 1)]


def workload(kind, mode, mt, n=1500, seed=0):
    rng = np.random.default_rng(500 + 10 * mode + mt + seed)
    model = mp.make_model(kind, 4)
    params = mp.default_params(model, mode, mt)
    J = model.n_joint
    boxes = common.random_boxes(rng, 30, [-2, -2, -1], [3, 2, 1], 0.05, 0.35, "so3")
    rows = []
    if mode != 0:
        if mt == 0:
            rows += [rng.uniform(-2, 3, n), rng.uniform(-2, 2, n), rng.uniform(-np.pi, np.pi, n)]
        else:
            rows += list(rng.uniform(-1.5, 1.5, (3, n))) + list(common.random_unit_quats(rng, n))
    if mode != 1:
        rows += list(rng.uniform(-1.58, 1.58, (J, n)))
    return model, params, boxes, np.array(rows)


def with_gates(params, stab, coll):
    p = A.Params.from_buffer_copy(params)
    p.check_stability, p.check_collision = stab, coll
    return p


@pytest.mark.parametrize("kind,mode,mt", COMBOS)
def test_oracle_gates_compose_from_the_two_stages(kind, mode, mt):
    model, params, boxes, s = workload(kind, mode, mt, n=600)
    both = O.eval_batch(model, boxes, with_gates(params, 1, 1), s, A.LAYOUT_OMPL, threads=4)
    for stab in (0, 1):
        for coll in (0, 1):
            r = O.eval_batch(model, boxes, with_gates(params, stab, coll), s, A.LAYOUT_OMPL, threads=4)
            want = (both["stable"].astype(bool) | (stab == 0)) & (~both["collision"].astype(bool) | (coll == 0))
            assert np.array_equal(r["flags"].astype(bool), want), (stab, coll)
            assert np.all(np.isinf(r["min_dist"])) if coll == 0 else np.array_equal(r["min_dist"], both["min_dist"])
    # -1 = the reference's own gating (motion_planning.cpp:138-153)
    ref_stab = 1 if (mode != 1 or mt == 1) else 0
    ref_coll = 1 if mode != 0 else 0
    auto = O.eval_batch(model, boxes, params, s, A.LAYOUT_OMPL, threads=4)
    forced = O.eval_batch(model, boxes, with_gates(params, ref_stab, ref_coll), s, A.LAYOUT_OMPL, threads=4)
    assert np.array_equal(auto["flags"], forced["flags"]) and np.array_equal(auto["band"], forced["band"])


@pytest.mark.gpu
@pytest.mark.parametrize("kind,mode,mt", COMBOS)
def test_gpu_gate_overrides_vs_oracle(kind, mode, mt):
    model, params, boxes, s = workload(kind, mode, mt)
    eng = mp.Engine(model, boxes, params)
    for stab in (0, 1):
        for coll in (0, 1):
            p = with_gates(params, stab, coll)
            ref = O.eval_batch(model, boxes, p, s, A.LAYOUT_OMPL, threads=8)
            eng.set_params(p)
            fast = eng.is_valid_batch(s, A.LAYOUT_OMPL)["flags"]                                    # flag path
            full = eng.is_valid_batch(s, A.LAYOUT_OMPL, want=("flags", "min_dist", "margin", "band"))  # full path
            band = (ref["band"] | full["band"]).astype(bool)
            assert not np.any((fast != ref["flags"]) & ~band), (stab, coll)
            assert not np.any((full["flags"] != ref["flags"]) & ~band), (stab, coll)
            fin = np.isfinite(ref["min_dist"])
            assert np.array_equal(fin, np.isfinite(full["min_dist"]))
            assert np.allclose(full["min_dist"][fin], ref["min_dist"][fin], rtol=1e-5, atol=1e-9)
            if coll == 0:
                assert not fin.any()         # collision stage off: +inf everywhere
            if stab == 0 and coll == 0:
                assert fast.all() and full["flags"].all()
    # a10: checkCollision(const MultilinkState&) = (0, 1), read through mvk_is_valid's result
    p = with_gates(params, 0, 1)
    eng.set_params(p)
    ref = O.eval_batch(model, boxes, p, s, A.LAYOUT_OMPL, threads=8)
    hits = 0
    for i in range(40):
        r = eng.is_valid(s[:, i])
        if not ref["band"][i]:
            assert r.collision == ref["collision"][i] and r.valid == (0 if ref["collision"][i] else 1)
            hits += int(r.collision)
        assert r.min_dist == pytest.approx(ref["min_dist"][i], rel=1e-5, abs=1e-9)
    eng.close()
    # edges honour the same gates
    rng = np.random.default_rng(3)
    E = 64
    a = s[:, :E].copy()
    b = a + 0.1 * (s[:, E:2 * E] - a)
    if mode != 0 and mt == 1:
        b[3:7] /= np.linalg.norm(b[3:7], axis=0)
    for stab, coll in ((0, 1), (1, 0)):
        p = with_gates(params, stab, coll)
        g_ok, g_first, g_band = O.check_motion_batch(model, boxes, p, a, b, 12)
        eng = mp.Engine(model, boxes, p)
        ok, first = eng.check_motion_batch(a, b, 12)
        eng.close()
        clear = ~g_band.astype(bool)
        assert np.array_equal(ok[clear], g_ok[clear]) and np.array_equal(first[clear], g_first[clear]), (stab, coll)
