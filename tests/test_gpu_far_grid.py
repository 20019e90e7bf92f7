"""Far grid behind the two distance grids (mvk_grid.cpp build_far_grid): primitives farther from every box than the
second grid's slack -- open space, the rim of the obstacles' volume -- take their cell's nearest-candidate list
instead of the sweep over all boxes.  MVK_NO_FAR_GRID=1 keeps the sweep; both collect a superset of the boxes that
can realise a minimum and take the minimum of the same per-pair values, so every output must carry the same bits --
in the lane-per-state kernels and the wave-per-state ones, full path and avoidance rows -- and both must agree
with the oracle.

Reference: the distance queries of CollisionAvoidance::getConstraint (collision_avoidance.cpp:147-153) and the
minimum distance of MotionPlanning::addState's statistics (motion_planning.cpp:546-642)."""
import numpy as np
import pytest

import common
import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A

pytestmark = pytest.mark.gpu


def _workloads():
    # the rim of the c2k volume and the space around it: most primitives beyond the second grid's radius, inside the
    # far grid's volume; the last third far outside it (the sweep stays)
    w = common.config2k(n=6144)
    st = w.soa.copy()
    n = st.shape[1]
    st[0] = np.linspace(-5.0, 9.0, n)
    st[1] *= 2.0
    st[2] = np.where(np.arange(n) % 3 == 0, 3.0, np.where(np.arange(n) % 3 == 1, -1.0, 12.0))
    yield "c2k rim", w.model, w.boxes, w.params, st, w.layout
    w = common.config3(n=6144)
    st = w.soa.copy()
    st[0:3] *= 3.5          # DRAGON around and beyond the [-3, 3]^3 scene
    yield "c3 rim", w.model, w.boxes, w.params, st, w.layout
    w = common.config5(n=6144)
    st = w.soa.copy()
    st[2] += np.linspace(0.0, 5.0, st.shape[1])
    yield "c5 rising", w.model, w.boxes, w.params, st, w.layout


def _same(x, y):
    return np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)


@pytest.mark.parametrize("name,model,boxes,params,soa,layout", list(_workloads()), ids=lambda v: v if isinstance(v, str) else "")
def test_far_grid_lists_give_the_bits_of_the_sweep(monkeypatch, name, model, boxes, params, soa, layout):
    sizes = (1, 37, 300, soa.shape[1])       # wave-per-state kernels up to 4096 states, lane-per-state beyond
    full, rows, stats = {}, {}, {}
    for far in (True, False):
        if not far:
            monkeypatch.setenv("MVK_NO_FAR_GRID", "1")
        eng = mp.Engine(model, boxes, params)
        full[far] = [eng.eval_batch(soa[:, :n], layout) for n in sizes]
        stats[far] = eng.reduce_stats()
        rows[far] = [eng.collision_avoidance_batch(soa[:, :n], layout) for n in sizes]
        eng.close()
    for a, b, n in zip(full[True], full[False], sizes):
        for key in ("flags", "band", "min_dist", "margin"):
            assert _same(a[key], b[key]), (name, n, key, np.flatnonzero(a[key] != b[key])[:5])
    assert stats[True] == stats[False]
    for a, b, n in zip(rows[True], rows[False], sizes):
        for key in ("dist", "box", "p_env", "p_robot", "row", "lb", "ok"):
            assert _same(a[key], b[key]), (name, n, key, np.argwhere(a[key] != b[key])[:5])
    # and against the oracle, on an evenly spaced subset of the batch
    idx = np.linspace(0, soa.shape[1] - 1, 1500).astype(int)
    sub = np.ascontiguousarray(soa[:, idx])
    ref = O.eval_batch(model, boxes, params, sub, layout, threads=8)
    got = full[True][-1]
    fin = np.isfinite(ref["min_dist"])
    assert np.allclose(got["min_dist"][idx][fin], ref["min_dist"][fin], rtol=0, atol=1e-9)
    assert (ref["min_dist"][fin] > 0.8).mean() > 0.1          # the workload is what it says: far from every box
    ra = O.avoid_batch(model, boxes, params, sub)
    ga = rows[True][-1]
    assert np.allclose(ga["dist"][:, idx], ra["dist"], rtol=1e-5, atol=1e-9)
    # (two boxes at the same penetration depth to the last bit may swap: a contact's depth is a face offset)
    assert (ga["box"][:, idx] == ra["box"]).mean() > 0.999
    assert np.array_equal(ga["ok"][idx], ra["ok"])


def test_new_scene_rebuilds_the_far_grid():
    w = common.config2k(n=512)
    st = w.soa.copy()
    st[2] += 3.0
    eng = mp.Engine(w.model, w.boxes, w.params)
    a = eng.collision_avoidance_batch(st, w.layout)
    moved = w.boxes.copy()
    moved[:, 2] += 2.0
    eng.set_scene(moved)
    b = eng.collision_avoidance_batch(st, w.layout)
    eng.set_scene(w.boxes)
    c = eng.collision_avoidance_batch(st, w.layout)
    eng.close()
    assert _same(a["dist"], c["dist"]) and not _same(a["dist"], b["dist"])
    ref = O.avoid_batch(w.model, moved, w.params, st)
    assert np.allclose(b["dist"], ref["dist"], rtol=1e-5, atol=1e-9)
