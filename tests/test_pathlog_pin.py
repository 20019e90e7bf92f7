"""PIN: the reference's own planner log pins OMPL's validSegmentCount + interpolate for the path.

tests/golden/ref_hydrus_planning_log.txt is a data file of the reference
(aerial_motion/sampling_based_method/path_log/hydrus/hydrus_planning_log.txt): 235 states written by
MotionPlanning::plan's densification loop (motion_planning.cpp:424-446) in mode 2 / SE2.  Way-points
appear twice in a row (addState(state2) then addState(state1) of the next pair), which splits the log
into 9 edges; each edge must have exactly validSegmentCount(s1, s2) segments (OMPL defaults 1 % / factor 1,
quirk Q4) and its interior states must be interpolate(s1, s2, j/nd) to the 6 significant digits printed.
"""
import math
import os

import numpy as np
import pytest

import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A

LOG = os.path.join(os.path.dirname(__file__), "golden", "ref_hydrus_planning_log.txt")


def load_states():
    rows = []
    for line in open(LOG):
        if line.startswith("state") and not line.startswith("states"):
            rows.append([float(x) for x in line.split()[1:]])
    return np.array(rows)


def edges_of(S):
    dup = [i for i in range(1, len(S)) if np.allclose(S[i], S[i - 1], atol=0, rtol=0)]
    out, start = [], 0
    for d in dup + [len(S)]:
        out.append((start, d - 1))
        start = d
    return out


def to_ompl(row):
    yaw = 2.0 * math.atan2(row[5], row[6])
    yaw = (yaw + math.pi) % (2 * math.pi) - math.pi
    return np.array([row[0], row[1], yaw, *row[7:10]])


@pytest.fixture(scope="module")
def spaces():
    m = mp.make_model("hydrus", 4)
    p = mp.default_params(m, A.MODE_JOINTS_AND_BASE, A.SE2)
    return mp.Space(p, 3), O.Space(p, 3)


def test_log_shape():
    S = load_states()
    assert S.shape == (235, 10)
    assert [b - a for a, b in edges_of(S)] == [33, 18, 9, 10, 14, 17, 38, 54, 33]


@pytest.mark.parametrize("which", ["product", "oracle"])
def test_segment_counts_and_interpolation_match_the_reference_log(spaces, which):
    sp = spaces[0] if which == "product" else spaces[1]
    S = load_states()
    for a, b in edges_of(S):
        s1, s2 = to_ompl(S[a]), to_ompl(S[b])
        nd = sp.valid_segment_count(s1, s2)
        assert nd == b - a, f"edge {a}-{b}: validSegmentCount {nd} != {b - a} states in the log"
        for j in range(1, nd):
            got = sp.unpack(sp.interpolate(s1, s2, j / nd))
            ref = S[a + j]
            # the log prints 6 significant digits
            tol = 5e-6 * np.maximum(1.0, np.abs(ref)) + 2e-6
            if np.dot(got[3:7], ref[3:7]) < 0:
                got[3:7] = -got[3:7]
            assert np.all(np.abs(got - ref) <= tol), (a, b, j, got, ref)


def test_product_space_equals_oracle_space_on_random_states():
    rng = np.random.default_rng(7)
    for kind, n_links, mode, mt in (("hydrus", 6, 2, 0), ("hydrus", 4, 1, 0), ("hydrus", 4, 0, 0), ("dragon", 4, 2, 1),
                                    ("dragon", 4, 1, 1)):
        m = mp.make_model(kind, n_links)
        p = mp.default_params(m, mode, mt)
        J = m.n_joint
        sp, so = mp.Space(p, J), O.Space(p, J)
        assert sp.dim == so.dim
        assert sp.max_extent() == pytest.approx(so.max_extent(), rel=1e-15)

        def sample():
            parts = []
            if mode != 0:
                if mt == 0:
                    parts += [rng.uniform(-2, 5), rng.uniform(-2.2, 2.2), rng.uniform(-np.pi, np.pi)]
                else:
                    q = rng.normal(size=4)
                    parts += [*rng.uniform(-2, 2, 3), *(q / np.linalg.norm(q))]
            if mode != 1:
                parts += list(rng.uniform(-1.58, 1.58, J))
            return np.array(parts)
        for _ in range(200):
            a, b, t = sample(), sample(), rng.uniform()
            assert sp.distance(a, b) == pytest.approx(so.distance(a, b), rel=1e-14, abs=1e-15)
            assert sp.valid_segment_count(a, b) == so.valid_segment_count(a, b)
            assert np.allclose(sp.interpolate(a, b, t), so.interpolate(a, b, t), rtol=1e-14, atol=1e-15)
            assert np.allclose(sp.unpack(a), so.unpack(a), rtol=0, atol=0)


def test_so2_wraps_and_so3_takes_the_short_way():
    m = mp.make_model("hydrus", 4)
    p = mp.default_params(m, A.MODE_ONLY_BASE, A.SE2)
    sp = mp.Space(p, 3)
    a, b = np.array([0.0, 0.0, 3.0]), np.array([0.0, 0.0, -3.0])
    assert sp.distance(a, b) == pytest.approx(0.5 * (2 * math.pi - 6.0))   # SO2 weight 0.5
    mid = sp.interpolate(a, b, 0.5)
    assert abs(abs(mid[2]) - math.pi) < 1e-12                               # through +-pi, not through 0
    p3 = mp.default_params(m, A.MODE_ONLY_BASE, A.SE3)
    s3 = mp.Space(p3, 3)
    qa = np.array([0, 0, 0, 0, 0, 0, 1.0])
    qb = np.array([0, 0, 0, 0, 0, -math.sin(0.4), -math.cos(0.4)])         # same rotation class as yaw 0.8
    mid = s3.interpolate(qa, qb, 0.5)
    assert np.allclose(np.abs(mid[3:]), [0, 0, math.sin(0.2), math.cos(0.2)], atol=1e-12)
    assert s3.distance(qa, qb) == pytest.approx(0.4)
