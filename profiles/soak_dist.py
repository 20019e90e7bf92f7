#!/usr/bin/env python3
"""The distance kernels' broadphase machinery on the fuzz test's random scenes, under many seeds: the default build of a
scene (distance grids sorted by bound, far grid, primitive groups) against the plain sweep over all boxes with one
primitive at a time (MVK_NO_DISTANCE_GRID=1 MVK_NO_PRIM_GROUPS=1) -- every output of the full path and of the
avoidance rows must carry the same bits, in both kernel sizes (lane per state, wave per state).
usage: python profiles/soak_dist.py [first] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
import motion_planning_amd as mp
from motion_planning_amd import _abi as A

DRAWS = [("hydrus", 4, 70), ("hydrus", 6, 300), ("dragon", 4, 1000), ("hydrus", 8, 600), ("dragon", 4, 120), ("hydrus", 6, 1000)]
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 6144
t0 = time.time()


def same(a, b):
    return np.array_equal(a, b, equal_nan=True) if a.dtype.kind == "f" else np.array_equal(a, b)


for off in range(first, first + count):
    for kind, n_links, n_boxes in DRAWS:
        rng = np.random.default_rng(55 + 1000 * n_links + n_boxes + (7 if kind == "dragon" else 0) + 100003 * off)
        model = mp.make_model(kind, n_links)
        se3 = kind == "dragon" or n_boxes % 2 == 0
        params = mp.default_params(model, A.MODE_JOINTS_AND_BASE, A.SE3 if se3 else A.SE2)
        J = model.n_joint
        boxes = common.random_boxes(rng, n_boxes, [-3, -3, -2], [3, 3, 2], 0.02, 0.3, "so3" if se3 else "yaw")
        boxes[0:10, 7] = 0.0
        boxes[10:20, 8:10] = 1e-3
        boxes[20, 7:10] = 0.0
        st = np.zeros((7 + J, n))
        scale = np.where(np.arange(n) % 4 == 3, 3.0, 1.0)
        st[0:3] = rng.uniform(-2.5, 2.5, (3, n)) * scale
        st[3:7] = common.random_unit_quats(rng, n) if se3 else common.yaw_quats(rng.uniform(-np.pi, np.pi, n))
        st[7:] = rng.uniform(-1.5, 1.5, (J, n))
        what = f"offset {off} {kind}-{n_links} {n_boxes} boxes"
        out = {}
        for mode in ("default", "sweep"):
            for k in ("MVK_NO_DISTANCE_GRID", "MVK_NO_PRIM_GROUPS"):
                if mode == "sweep": os.environ[k] = "1"
                else: os.environ.pop(k, None)
            eng = mp.Engine(model, boxes, params)
            out[mode] = (eng.eval_batch(st, A.LAYOUT_POSE7), eng.eval_batch(st[:, :300], A.LAYOUT_POSE7),
                         eng.collision_avoidance_batch(st[:, :4608], A.LAYOUT_POSE7), eng.collision_avoidance_batch(st[:, :200], A.LAYOUT_POSE7))
            eng.close()
        for k in ("MVK_NO_DISTANCE_GRID", "MVK_NO_PRIM_GROUPS"):
            os.environ.pop(k, None)
        for i, keys in ((0, ("flags", "band", "min_dist", "margin")), (1, ("flags", "band", "min_dist", "margin")),
                        (2, ("dist", "box", "p_env", "p_robot", "row", "lb", "ok")), (3, ("dist", "box", "p_env", "p_robot", "row", "lb", "ok"))):
            for key in keys:
                a, b = out["default"][i][key], out["sweep"][i][key]
                if not same(a, b):
                    idx = np.argwhere(~((a == b) | ((a != a) & (b != b))))[:4] if a.dtype.kind == "f" else np.argwhere(a != b)[:4]
                    print(f"FAILED {what}: output set {i} key {key} differs at {idx.tolist()}: {a[tuple(idx[0])]!r} vs {b[tuple(idx[0])]!r}", flush=True)
                    sys.exit(1)
    print(f"seed offset {off}: {len(DRAWS)} scenes ok ({time.time() - t0:.0f} s)", flush=True)
print(f"{count} offsets x {len(DRAWS)} scenes: grids + groups give the bits of the plain sweep, full path and avoidance rows, both kernel sizes")
