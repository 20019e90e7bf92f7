"""Primitive groups in the full path's distance kernel and the avoidance rows' kernel (mvk_capi.cpp derive_extras,
mvk_kernels.hip collide_full / avoid_kernel): consecutive primitives of one segment whose anchors lie within a few
centimetres of each other (a link's bar and its propeller protector; a link's pipe and its gimbal modules) share one
broadphase query about the leader's centre with a radius that
contains every member; each member then runs its own narrowphase over the shared candidate list.  The list is a
superset of every member's own list and the per-pair distances are the same code, so every output must carry the bits
of the ungrouped kernel (MVK_NO_PRIM_GROUPS=1) -- and both must agree with the oracle.

Reference: the per-link FCL distance requests of MotionPlanning::isStateValid (motion_planning.cpp:96-164) and the
minimum distance of addState's statistics (motion_planning.cpp:546-642)."""
import numpy as np
import pytest

import common
import motion_planning_amd as mp
import oracle_lib as O

pytestmark = pytest.mark.gpu


def _workloads():
    for name, make in (("c2k", common.config2k), ("c2", common.config2), ("c3", common.config3), ("c5", common.config5)):
        w = make(n=12288)
        yield name, w.model, w.boxes, w.params, w.soa, w.layout
    # the rim of the scenes and the space around them: the group walks then go through the far grid (whose lists must be
    # complete for the groups' radius term, grp_rq, not only for a single primitive's) and the sweep over all boxes
    from test_gpu_far_grid import _workloads as rim
    yield from rim()


def _same(x, y):
    return np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)


@pytest.mark.parametrize("name,model,boxes,params,soa,layout", list(_workloads()), ids=lambda v: v if isinstance(v, str) else "")
def test_grouped_broadphase_gives_the_bits_of_the_ungrouped_one(monkeypatch, name, model, boxes, params, soa, layout):
    sizes = (1, 300, soa.shape[1])       # wave-per-state kernels up to 4096 states, lane-per-state beyond
    full, stats, rows = {}, {}, {}
    for grouped in (True, False):
        if not grouped:
            monkeypatch.setenv("MVK_NO_PRIM_GROUPS", "1")
        eng = mp.Engine(model, boxes, params)
        full[grouped] = [eng.eval_batch(soa[:, :n], layout) for n in sizes]
        stats[grouped] = eng.reduce_stats()
        rows[grouped] = [eng.collision_avoidance_batch(soa[:, :n], layout) for n in sizes]
        eng.close()
    for a, b, n in zip(full[True], full[False], sizes):
        for key in ("flags", "band", "min_dist", "margin"):
            assert _same(a[key], b[key]), (name, n, key, np.flatnonzero(a[key] != b[key])[:5])
    assert stats[True] == stats[False]
    # the avoidance rows share one cell walk per group as well (avoid_kernel), every member draining the group's list
    for a, b, n in zip(rows[True], rows[False], sizes):
        for key in ("dist", "box", "p_env", "p_robot", "row", "lb", "ok"):
            assert _same(a[key], b[key]), (name, n, key, np.argwhere(a[key] != b[key])[:5])
    idx = np.linspace(0, soa.shape[1] - 1, 1200).astype(int)
    sub = np.ascontiguousarray(soa[:, idx])
    ref = O.eval_batch(model, boxes, params, sub, layout, threads=8)
    got = full[True][-1]
    fin = np.isfinite(ref["min_dist"])
    assert np.array_equal(got["flags"][idx], ref["flags"])
    assert np.allclose(got["min_dist"][idx][fin], ref["min_dist"][fin], rtol=0, atol=1e-9)
