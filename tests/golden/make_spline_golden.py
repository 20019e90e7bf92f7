"""Golden vectors for the batched B-spline evaluation (mvk_spline_eval[_dev]): values of scipy.interpolate.BSpline --
an independent implementation of the clamped uniform B-spline tinyspline builds for BsplineRos::initialize
(bspline_ros/src/bspline_ros.cpp:45-86) -- over the control points ContinuousPathGenerator::calcContinuousPath
forms (continuous_path_generator.cpp:240-271: cog xyz, roll pitch yaw with a continuous yaw, joints; first and last
state doubled).  Run here (scipy present), commit the .npz: the GPU box then needs nothing but the repo.

usage: python tests/golden/make_spline_golden.py
"""
import os
import sys

import numpy as np
from scipy.interpolate import BSpline

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from test_path_post import _random_path  # noqa: E402


def rpy(q):
    x, y, z, w = q / np.linalg.norm(q)
    r20, r21, r22 = 2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)
    r10, r00 = 2 * (x * y + w * z), 1 - 2 * (y * y + z * z)
    pitch = -np.arcsin(r20)
    return np.arctan2(r21 / np.cos(pitch), r22 / np.cos(pitch)), pitch, np.arctan2(r10 / np.cos(pitch), r00 / np.cos(pitch))


def control_points(path):
    n = len(path)
    ctrl, prev = [], None
    for i in [0] + list(range(n)) + [n - 1]:
        r, p, y = rpy(path[i, 3:7])
        if prev is None:
            prev = y
        while y - prev > np.pi:
            y -= 2 * np.pi
        while y - prev < -np.pi:
            y += 2 * np.pi
        prev = y
        ctrl.append([*path[i, 0:3], r, p, y, *path[i, 7:]])
    return np.array(ctrl)


def main():
    for degree, J, n, nt, period in ((5, 5, 40, 1501, 100.0), (3, 3, 12, 257, 7.5)):
        rng = np.random.default_rng(100 + degree)
        path = _random_path(rng, n, J)
        ctrl = control_points(path)
        nc = len(ctrl)
        knots = np.concatenate([np.zeros(degree + 1), np.arange(1, nc - degree) / (nc - degree), np.ones(degree + 1)])
        ref = BSpline(knots, ctrl, degree)
        t = np.linspace(-0.5, period + 0.5, nt)
        x = np.clip(t, 0, period) / period
        vals = [ref(x)] + [ref.derivative(k)(x) / period ** k for k in (1, 2)]
        np.savez_compressed(os.path.join(HERE, f"spline_d{degree}.npz"), path=path, period=period, degree=degree, t=t,
                            d0=vals[0], d1=vals[1], d2=vals[2])
        print(f"spline_d{degree}.npz: {nc} control points, {nt} times")


if __name__ == "__main__":
    main()
