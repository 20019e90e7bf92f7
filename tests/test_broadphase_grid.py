"""Host tests of the broadphase grid (mvk_grid.cpp) through the C ABI's diagnostic entry point.

The grid replaces FCL's broadphase under PlanningScene::checkCollision (call site motion_planning.cpp:156-161).
Property that keeps the flags exact: for any query point, every box within `radius` of it is listed by its cell.
Distances to oriented boxes are recomputed here in numpy (closed form: clamp in the box frame).
"""
import numpy as np
import pytest

import common
import motion_planning_amd as mp


def point_box_dist(points, boxes):
    """[n, m] exact distance from points [n,3] to oriented boxes [m,10]"""
    c, q, half = boxes[:, 0:3], boxes[:, 3:7], 0.5 * boxes[:, 7:10]
    x, y, z, w = q.T / np.linalg.norm(q, axis=1)
    R = np.stack([np.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y)], -1),
                  np.stack([2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x)], -1),
                  np.stack([2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], -1)], 1)   # [m,3,3]
    d = points[:, None, :] - c[None, :, :]                    # [n,m,3]
    t = np.einsum("mki,nmk->nmi", R, d)                       # R^T d
    e = np.maximum(np.abs(t) - half[None], 0.0)
    return np.sqrt((e * e).sum(-1))


@pytest.mark.parametrize("case", ["c2", "c2k", "c3", "gap_h", "gap_v", "one", "flat"])
def test_cell_lists_are_conservative_and_tight(case):
    rng = np.random.default_rng(5)
    radius = 0.33
    if case == "c2":
        boxes = common.config2(n=1).boxes
    elif case == "c2k":
        boxes = common.config2k(n=1).boxes
    elif case == "c3":
        boxes = common.config3(n=1).boxes
        radius = 0.26
    elif case == "gap_h":
        boxes = mp.horizontal_gap_scene()
    elif case == "gap_v":
        boxes = mp.vertical_gap_scene(0.5, 0.5, 0.7, True)
    elif case == "one":
        boxes = np.array([[0.3, -0.2, 0.1, 0.1, 0.2, 0.3, 0.9, 0.4, 0.2, 0.6]])
    else:   # zero-thickness plates
        boxes = common.random_boxes(rng, 20, [-1] * 3, [1] * 3, 0.0, 0.3, "so3")
        boxes[:, 7] = 0.0
    lo = boxes[:, 0:3].min(0) - 1.5
    hi = boxes[:, 0:3].max(0) + 1.5
    if case in ("gap_h", "gap_v"):
        lo, hi = np.array([-6.0, -6.0, -6.0]), np.array([6.0, 6.0, 6.0])
    n = 4000
    pts = rng.uniform(lo, hi, (n, 3))
    # a third of the points hugging box surfaces, where a missed neighbour would matter
    k = rng.integers(0, len(boxes), n // 3)
    pts[: n // 3] = boxes[k, 0:3] + rng.normal(size=(n // 3, 3)) * (0.5 * boxes[k, 7:10].max(1, keepdims=True) + radius)
    counts, ids = mp.broadphase_query(boxes, radius, pts, cap=1024)
    assert counts.max() <= 1024
    dist = point_box_dist(pts, boxes)
    near = dist <= radius
    listed = np.zeros_like(near)
    rows = np.repeat(np.arange(n), ids.shape[1])
    flat = ids.ravel()
    ok = flat >= 0
    listed[rows[ok], flat[ok]] = True
    assert not np.any(near & ~listed), "a box within the radius is missing from its cell's list"
    # tight enough to be useful: nothing farther than radius + 2 cell diagonals is ever listed
    far = dist[listed].max() if listed.any() else 0.0
    assert far < radius + 2.5, far
    if case in ("c2k", "c3"):
        assert listed.sum() < 0.1 * near.size      # the lists cut the pair count by an order of magnitude
        assert listed.sum() < 12 * max(near.sum(), 1)


def test_empty_scene_and_capacity():
    counts, ids = mp.broadphase_query(np.zeros((0, 10)), 0.3, np.zeros((5, 3)), cap=4)
    assert np.all(counts == 0)
    too_many = np.zeros((65535, 10))
    too_many[:, 6] = 1.0
    with pytest.raises(mp.MvkError):
        mp.broadphase_query(too_many, 0.3, np.zeros((1, 3)))
