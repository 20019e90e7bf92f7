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


def test_path_log_load_reads_the_reference_savepath_format(tmp_path):
    """savePath of the reference's current code prints J newlines after every state (motion_planning.cpp:682-684,
    quirk Q6): J - 1 blank lines between states.  ADVICE r1: such a file must load."""
    lines = open(LOG).read().splitlines()
    out = []
    for ln in lines:
        out.append(ln)
        if ln.startswith("state") and not ln.startswith("states"):
            out += ["", ""]          # J = 3: two blank lines
    f = tmp_path / "q6.txt"
    f.write_text("\n".join(out) + "\n")
    hdr, states = PL.load_path_log(f, 3)
    assert hdr["n_states"] == 235 and np.array_equal(states, load_states())
    # the mode-1 log of the reference (ten values per state, J = 3) loads too
    hdr1, st1 = PL.load_path_log(os.path.join(os.path.dirname(LOG), "ref_hydrus_planning_log_mode1.txt"), 3)
    assert hdr1["n_states"] == 390 and hdr1["planning_mode"] == 1 and st1.shape == (390, 10)
    assert hdr1["planning_time"] == pytest.approx(0.208125)


def test_checker_with_half_a_split_interface_is_rejected():
    """begin without end used to crash in check_end"""
    import ctypes as C
    model, params, boxes, start, goal = common.config1()
    lib = A.load()
    noop_s = A.CHECK_STATES_FN(lambda *a: 0)
    noop_m = A.CHECK_MOTIONS_FN(lambda *a: 0)
    begin = C.cast(C.CFUNCTYPE(C.c_int)(lambda: 0), C.c_void_p)
    ck = A.Checker(noop_s, noop_m, None, begin, None)
    s = np.ascontiguousarray(start)
    g = np.ascontiguousarray(goal)
    req = A.PlanRequest(s.ctypes.data_as(C.POINTER(C.c_double)), g.ctypes.data_as(C.POINTER(C.c_double)), 0.0, 1.0, 1, 1, 4)
    wp = np.zeros((16, 6))
    res = A.PlanResult()
    rc = lib.mvk_plan_rrt_connect_cb(C.byref(params), 3, C.byref(ck), C.byref(req), wp.ctypes.data_as(C.c_void_p), 16, C.byref(res))
    assert rc == A.ERR_INVALID_ARG and b"both or neither" in lib.mvk_last_error()


_FAR_SCRIPT = r"""
import sys, json
import numpy as np
sys.path.insert(0, sys.argv[1]); sys.path.insert(0, sys.argv[1] + "/tests")
import common, oracle_lib as O
from motion_planning_amd import _abi as A, planner as PL
shift = float(sys.argv[2])
model, params, boxes, start, goal = common.config1()
boxes = boxes.copy(); boxes[:, 0] += shift
params.pos_low[0] += shift; params.pos_high[0] += shift
start[0] += shift; goal[0] += shift
cs = lambda soa: O.eval_batch(model, boxes, params, soa, A.LAYOUT_OMPL)["flags"]
def cm(s1, s2, nd):
    ok, first, _ = O.check_motion_batch(model, boxes, params, s1, s2, nd)
    return ok, first
wp, info = PL.plan_rrt_connect_cb(params, 3, cs, cm, start, goal, time_limit_s=120.0, seed=5, batch=16)
print(json.dumps(dict(iterations=int(info["iterations"]), tree=list(map(int, info["tree_sizes"])), wp=np.asarray(wp).tolist())))
"""


@pytest.mark.parametrize("shift", [0.0, 3000.0])
def test_two_pass_nearest_search_equals_the_one_pass_scan_far_from_the_origin(shift, tmp_path):
    """the fp32 pre-pass of the nearest-node search keeps the nodes within a slack of the fp32 minimum; the
    slack now follows from the space's bounds, so the fp64 nearest node is among them wherever the scene sits (with
    the fixed 2e-4 of round 2 a scene 3 km from the origin -- fp32 spacing 2.4e-4 m -- could pick another node).
    Same seed => same plan as the fp64 one-pass scan (MVK_PLAN_ONE_PASS, read once per process: two children)."""
    import json
    import subprocess
    import sys
    script = tmp_path / "far.py"
    script.write_text(_FAR_SCRIPT)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = []
    for one_pass in (False, True):
        env = dict(os.environ)
        env.pop("MVK_PLAN_ONE_PASS", None)
        if one_pass:
            env["MVK_PLAN_ONE_PASS"] = "1"
        r = subprocess.run([sys.executable, str(script), root, str(shift)], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        out.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert out[0]["iterations"] > 3 and len(out[0]["wp"]) >= 2
    assert out[0] == out[1]
