"""ctypes plumbing for the host-side planner entry points of include/mvk.h (RRT-Connect, densification,
reference-format path logs).  The search itself is C++ (csrc/mvk_planner.cpp)."""
import ctypes as C

import numpy as np

from . import _abi as A
from .engine import MvkError, _check, _ptr


def _request(start, goal, rng_range, time_limit_s, seed, max_iterations, batch):
    start = np.ascontiguousarray(start, dtype=np.float64)
    goal = np.ascontiguousarray(goal, dtype=np.float64)
    req = A.PlanRequest(start.ctypes.data_as(C.POINTER(C.c_double)), goal.ctypes.data_as(C.POINTER(C.c_double)),
                        float(rng_range), float(time_limit_s), int(seed), int(max_iterations), int(batch))
    return req, (start, goal)


def _finish(lib, rc, res, wp, dim):
    info = {k: getattr(res, k) for k, _ in A.PlanResult._fields_ if k != "tree_sizes"}
    info["tree_sizes"] = list(res.tree_sizes)
    if rc == A.ERR_NO_SOLUTION:
        info["error"] = lib.mvk_last_error().decode()
        return None, info
    _check(lib, rc)
    return wp[: res.n_waypoints].copy(), info


def plan_rrt_connect(engine, start, goal, range_=0.0, time_limit_s=10.0, seed=1, max_iterations=0, batch=16, cap=4096):
    """MotionPlanning::plan on the GPU checker.  Returns (waypoints [k, dim] or None, info dict)."""
    lib = A.load()
    dim = engine.state_dim(A.LAYOUT_OMPL)
    req, keep = _request(start, goal, range_, time_limit_s, seed, max_iterations, batch)
    wp = np.zeros((cap, dim))
    res = A.PlanResult()
    rc = lib.mvk_plan_rrt_connect(engine.ctx, C.byref(req), _ptr(wp), cap, C.byref(res))
    return _finish(lib, rc, res, wp, dim)


def plan_rrt_connect_cb(params, n_joint, check_states, check_motions, start, goal, range_=0.0, time_limit_s=10.0,
                        seed=1, max_iterations=0, batch=16, cap=4096):
    """Same search with validity supplied by Python callables (used to run the host logic on a CPU checker):
    check_states(soa[dim, n]) -> uint8[n];  check_motions(s1[dim, E], s2[dim, E], nd[E]) -> (ok[E], first[E])."""
    lib = A.load()
    dim = lib.mvk_space_dim(C.byref(params), n_joint)

    def _cs(user, soa, n, ld, ok):
        arr = np.ctypeslib.as_array(soa, shape=(dim, ld))[:, :n]
        np.ctypeslib.as_array(ok, shape=(n,))[:] = check_states(np.ascontiguousarray(arr))
        return 0

    def _cm(user, s1, s2, n, ld, nd, ok, first):
        a = np.ascontiguousarray(np.ctypeslib.as_array(s1, shape=(dim, ld))[:, :n])
        b = np.ascontiguousarray(np.ctypeslib.as_array(s2, shape=(dim, ld))[:, :n])
        o, f = check_motions(a, b, np.ctypeslib.as_array(nd, shape=(n,)).copy())
        np.ctypeslib.as_array(ok, shape=(n,))[:] = o
        np.ctypeslib.as_array(first, shape=(n,))[:] = f
        return 0
    ck = A.Checker(A.CHECK_STATES_FN(_cs), A.CHECK_MOTIONS_FN(_cm), None)
    req, keep = _request(start, goal, range_, time_limit_s, seed, max_iterations, batch)
    wp = np.zeros((cap, dim))
    res = A.PlanResult()
    rc = lib.mvk_plan_rrt_connect_cb(C.byref(params), n_joint, C.byref(ck), C.byref(req), _ptr(wp), cap, C.byref(res))
    return _finish(lib, rc, res, wp, dim)


def densify(params, n_joint, waypoints):
    lib = A.load()
    wp = np.ascontiguousarray(waypoints, dtype=np.float64)
    n = lib.mvk_path_densify(C.byref(params), n_joint, _ptr(wp), len(wp), None, 0)
    if n < 0:
        _check(lib, n)
    out = np.zeros((n, wp.shape[1]))
    _check(lib, min(0, lib.mvk_path_densify(C.byref(params), n_joint, _ptr(wp), len(wp), _ptr(out), n)))
    return out


def save_path_log(file, states, n_joint, planning_mode, start, goal, planning_time=0.0, motion_cost=-1.0,
                  minimum_var=1e6, minimum_var_state_entry=0):
    lib = A.load()
    states = np.ascontiguousarray(states, dtype=np.float64)
    h = A.PathLog()
    h.n_states, h.n_joint, h.planning_mode = len(states), n_joint, planning_mode
    h.planning_time, h.motion_cost, h.minimum_var = planning_time, motion_cost, minimum_var
    h.minimum_var_state_entry = minimum_var_state_entry
    h.start[: 7 + n_joint] = list(start)
    h.goal[: 7 + n_joint] = list(goal)
    _check(lib, lib.mvk_path_log_save(str(file).encode(), C.byref(h), _ptr(states)))


def load_path_log(file, n_joint):
    lib = A.load()
    h = A.PathLog()
    n = lib.mvk_path_log_load(str(file).encode(), n_joint, C.byref(h), None, 0)
    if n < 0:
        _check(lib, n)
    states = np.zeros((n, 7 + n_joint))
    lib.mvk_path_log_load(str(file).encode(), n_joint, C.byref(h), _ptr(states), n)
    hdr = {k: getattr(h, k) for k in ("n_states", "planning_mode", "planning_time", "motion_cost", "minimum_var",
                                      "minimum_var_state_entry")}
    hdr["start"] = np.array(h.start[: 7 + n_joint])
    hdr["goal"] = np.array(h.goal[: 7 + n_joint])
    return hdr, states
