"""Host logic above the C ABI: RRT-Connect (driven here by the CPU oracle through the callback form --
config c1 "plumbing, no GPU"), densification pinned by the reference log, path-log I/O round trip."""
import os

import numpy as np
import pytest

import common
import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A
from motion_planning_amd import planner as PL
from test_pathlog_pin import LOG, edges_of, load_states, to_ompl


def oracle_checker(model, boxes, params):
    def cs(soa):
        return O.eval_batch(model, boxes, params, soa, A.LAYOUT_OMPL)["flags"]

    def cm(s1, s2, nd):
        ok, first, _ = O.check_motion_batch(model, boxes, params, s1, s2, nd)
        return ok, first
    return cs, cm


def test_densify_reproduces_the_reference_log():
    """way-points of the log -> mvk_path_densify -> the log's 235 states (duplicated way-points included)"""
    S = load_states()
    m = mp.make_model("hydrus", 4)
    p = mp.default_params(m, A.MODE_JOINTS_AND_BASE, A.SE2)
    ed = edges_of(S)
    wps = np.array([to_ompl(S[a]) for a, _ in ed] + [to_ompl(S[ed[-1][1]])])
    dense = PL.densify(p, 3, wps)
    assert len(dense) == 235
    sp = mp.Space(p, 3)
    for k in range(235):
        got = sp.unpack(dense[k])
        if np.dot(got[3:7], S[k][3:7]) < 0:
            got[3:7] = -got[3:7]
        assert np.all(np.abs(got - S[k]) <= 5e-6 * np.maximum(1.0, np.abs(S[k])) + 2e-6), k


def test_path_log_roundtrip_with_the_reference_file(tmp_path):
    hdr, states = PL.load_path_log(LOG, 3)
    assert hdr["n_states"] == 235 and hdr["planning_mode"] == 2
    assert hdr["planning_time"] == pytest.approx(112.726) and hdr["motion_cost"] == pytest.approx(7.37553)
    assert hdr["minimum_var"] == pytest.approx(0.452857) and hdr["minimum_var_state_entry"] == 200
    assert np.allclose(hdr["start"], [-1, 0, 0, 0, 0, -0.706825, 0.707388, 1.57, 1.57, 1.57])
    assert np.allclose(hdr["goal"], [1, 1, 0, 0, 0, -0.706825, 0.707388, 1.57, 1.57, 1.57])   # Q7 not reproduced
    assert np.array_equal(states, load_states())
    out = tmp_path / "log.txt"
    PL.save_path_log(out, states, 3, 2, hdr["start"], hdr["goal"], hdr["planning_time"], hdr["motion_cost"],
                     hdr["minimum_var"], hdr["minimum_var_state_entry"])
    # identical token stream to the reference file
    assert open(out).read().split() == open(LOG).read().split()
    with pytest.raises(mp.MvkError):
        PL.load_path_log(tmp_path / "missing.txt", 3)


def test_rrt_connect_gap_pass_on_the_cpu_checker():
    """c1: HYDRUS quad through the default horizontal gap world, CPU checker only"""
    model, params, boxes, start, goal = common.config1()
    cs, cm = oracle_checker(model, boxes, params)
    wp, info = PL.plan_rrt_connect_cb(params, 3, cs, cm, start, goal, time_limit_s=60.0, seed=3, batch=8)
    assert wp is not None, info
    assert info["solved"] == 1 and info["n_waypoints"] == len(wp) >= 2
    assert np.array_equal(wp[0], start) and np.array_equal(wp[-1], goal)
    # every edge of the answer is valid under the reference's own densification rule
    sp = O.Space(params, 3)
    nd = np.array([max(1, sp.valid_segment_count(wp[i], wp[i + 1])) for i in range(len(wp) - 1)], np.int32)
    ok, _, _ = O.check_motion_batch(model, boxes, params, wp[:-1].T.copy(), wp[1:].T.copy(), nd)
    assert ok.all()
    dense = PL.densify(params, 3, wp)
    flags = O.eval_batch(model, boxes, params, dense.T.copy(), A.LAYOUT_OMPL)["flags"]
    assert flags.all()
    assert info["path_length"] == pytest.approx(sum(sp.distance(wp[i], wp[i + 1]) for i in range(len(wp) - 1)))
    print("c1 CPU-checker plan:", info)


def test_rrt_connect_reports_invalid_start_and_no_solution():
    model, params, boxes, start, goal = common.config1()
    cs, cm = oracle_checker(model, boxes, params)
    bad = start.copy()
    bad[0:2] = [0.15, 1.0]      # inside the left wall
    wp, info = PL.plan_rrt_connect_cb(params, 3, cs, cm, bad, goal, time_limit_s=5.0)
    assert wp is None and "start state is invalid" in info["error"]
    # a sealed world: no solution within the iteration budget
    sealed = np.vstack([boxes, [[0.15, 0.0, 0, 0, 0, 0, 1, 0.3, 0.7, 10]]])
    cs, cm = oracle_checker(model, sealed, params)
    wp, info = PL.plan_rrt_connect_cb(params, 3, cs, cm, start, goal, time_limit_s=30.0, max_iterations=15, batch=4)
    assert wp is None and info["solved"] == 0 and info["iterations"] == 15
