"""Distance grids with bounds (mvk_grid.cpp build_grid, by_bound): a cell's list is sorted by the box-to-cell lower
bound and every group of four ids carries the bound of its first entry, so a kernel's walk over the list -- the nearest
box of a primitive's centre, then the boxes within a threshold of it -- stops at the first group beyond the threshold
instead of reading the whole list (mvk_kernels.hip dist_pop, grid_collect_nearest).

Host properties that keep the distances exact: the sorted list holds the boxes of the plain list; a group's bound is a
lower bound of the true distance from the query point to that group's boxes and to every later one, with the pad
(the device's fp32 slack) taken off; the bounds never decrease along a list.  On the GPU, MVK_NO_GRID_BOUNDS=1 (plain
lists, whole walks) gives every output the same bits.

Reference: the nearest-box queries of CollisionAvoidance::getConstraint (collision_avoidance.cpp:147-153) and of
MotionPlanning::addState's statistics (motion_planning.cpp:546-642)."""
import numpy as np
import pytest

import common
import motion_planning_amd as mp
from test_broadphase_grid import point_box_dist


@pytest.mark.parametrize("case", ["c2", "c2k", "c3", "plates"])
def test_group_bounds_are_lower_bounds_and_sorted(case):
    rng = np.random.default_rng(11)
    radius, pad = 0.9, 3e-5
    if case == "c2":
        boxes = common.config2(n=1).boxes
    elif case == "c2k":
        boxes = common.config2k(n=1).boxes
    elif case == "c3":
        boxes = common.config3(n=1).boxes
        radius = 1.0
    else:
        boxes = common.random_boxes(rng, 40, [-1] * 3, [1] * 3, 0.0, 0.3, "so3")
        boxes[:, 7] = 0.0
    n = 1500
    pts = rng.uniform(boxes[:, 0:3].min(0) - 1.0, boxes[:, 0:3].max(0) + 1.0, (n, 3))
    cap = 1024
    counts, ids, bounds = mp.grid_bounds_query(boxes, radius, pad, pts, cap=cap)
    plain_counts, plain_ids = mp.broadphase_query(boxes, radius, pts, cap=cap)
    assert counts.max() <= cap and np.array_equal(counts, plain_counts)
    assert np.array_equal(np.sort(ids, axis=1), np.sort(plain_ids, axis=1))         # the same boxes, another order
    dist = point_box_dist(pts, boxes)
    checked = 0
    for i in range(n):
        k = counts[i]
        if k == 0:
            continue
        b, d = bounds[i, :k].astype(np.float64), dist[i, ids[i, :k]]
        assert np.all(np.diff(b) >= 0.0)                                            # sorted by bound
        assert np.array_equal(b, b[4 * (np.arange(k) // 4)])                           # one bound per group of four
        suffix_min = np.minimum.accumulate(d[::-1])[::-1]                           # nearest of this entry and all later ones
        assert np.all(b <= np.maximum(suffix_min - pad, 0.0) + 1e-12), (i, np.max(b - suffix_min))   # (bounds are clamped at 0)
        checked += k
    assert checked > 3 * n
    # and useful: in the dense scenes most of a list lies beyond the nearest box's distance + a link's bounding radius
    if case in ("c2k", "c3"):
        near = dist.min(1) + 0.35
        beyond = sum(int((bounds[i, :counts[i]] > near[i]).sum()) for i in range(n))
        assert beyond > 0.25 * counts.sum(), (beyond, counts.sum())


def test_plain_grids_carry_no_bounds_and_bad_arguments_are_refused():
    boxes = common.config2(n=1).boxes
    with pytest.raises(mp.MvkError):
        mp.grid_bounds_query(boxes, -1.0, 0.0, np.zeros((1, 3)))
    with pytest.raises(mp.MvkError):
        mp.grid_bounds_query(boxes, 0.5, -1.0, np.zeros((1, 3)))
    counts, ids, bounds = mp.grid_bounds_query(np.zeros((0, 10)), 0.3, 0.0, np.zeros((5, 3)), cap=4)
    assert np.all(counts == 0) and np.all(ids == -1)


def _same(x, y):
    return np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["c2k", "c3", "c5"])
def test_early_stops_give_the_bits_of_whole_walks(monkeypatch, name):
    w = {"c2k": common.config2k, "c3": common.config3, "c5": common.config5}[name](n=8192)
    sizes = (1, 300, w.soa.shape[1])       # wave-per-state kernels up to 4096 states, lane-per-state beyond
    full, rows, stats = {}, {}, {}
    for bounds in (True, False):
        if not bounds:
            monkeypatch.setenv("MVK_NO_GRID_BOUNDS", "1")
        eng = mp.Engine(w.model, w.boxes, w.params)
        full[bounds] = [eng.eval_batch(w.soa[:, :n], w.layout) for n in sizes]
        stats[bounds] = eng.reduce_stats()
        rows[bounds] = [eng.collision_avoidance_batch(w.soa[:, :n], w.layout) for n in sizes]
        eng.close()
    for a, b, n in zip(full[True], full[False], sizes):
        for key in ("flags", "band", "min_dist", "margin"):
            assert _same(a[key], b[key]), (name, n, key, np.flatnonzero(a[key] != b[key])[:5])
    assert stats[True] == stats[False]
    for a, b, n in zip(rows[True], rows[False], sizes):
        for key in ("dist", "box", "p_env", "p_robot", "row", "lb", "ok"):
            assert _same(a[key], b[key]), (name, n, key, np.argwhere(a[key] != b[key])[:5])
