"""ctypes handle on oracle/_build/liboracle.so (the CPU checker).  Test infrastructure:
imported only from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os

import numpy as np

from motion_planning_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_LIB = None


class BatchOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("flags", "min_dist", "margin", "band", "stable", "collision", "fc_t_min",
                                           "fc_rp_min", "thrust", "rp_margin", "cog", "gimbal", "pivot_ratio")]


def load():
    global _LIB
    if _LIB is None:
        path = os.path.join(ROOT, "oracle", "_build", "liboracle.so")
        if not os.path.exists(path):
            raise RuntimeError(f"{path} missing: run `make -C oracle` (or __graft_entry__.build())")
        lib = C.CDLL(path)
        lib.orc_space_distance.restype = C.c_double
        lib.orc_space_max_extent.restype = C.c_double
        lib.orc_prim_box_signed_dist.restype = C.c_double
        _LIB = lib
    return _LIB


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def eval_batch(model, boxes, params, soa, layout=A.LAYOUT_POSE7, brute=False, threads=1):
    """boxes: [M,10] array.  Returns dict of per-sample arrays."""
    from motion_planning_amd.engine import boxes_from_array
    lib = load()
    soa = np.ascontiguousarray(soa, dtype=np.float64)
    rows, n = soa.shape
    N = model.n_rotor
    barr, m = boxes_from_array(boxes)
    out = dict(flags=np.zeros(n, np.uint8), min_dist=np.zeros(n), margin=np.zeros(n), band=np.zeros(n, np.uint8),
               stable=np.zeros(n, np.uint8), collision=np.zeros(n, np.uint8), fc_t_min=np.zeros(n),
               fc_rp_min=np.zeros(n), thrust=np.zeros((N, n)), rp_position_margin=np.zeros(n), cog=np.zeros((3, n)),
               gimbal=np.zeros((2 * N, n)), pivot_ratio=np.zeros(n))
    bo = BatchOut(_p(out["flags"]), _p(out["min_dist"]), _p(out["margin"]), _p(out["band"]), _p(out["stable"]),
                  _p(out["collision"]), _p(out["fc_t_min"]), _p(out["fc_rp_min"]), _p(out["thrust"]),
                  _p(out["rp_position_margin"]), _p(out["cog"]), _p(out["gimbal"]), _p(out["pivot_ratio"]))
    rc = lib.orc_eval_batch(C.byref(model), barr, C.c_size_t(m), C.byref(params), _p(soa), C.c_size_t(n),
                            C.c_size_t(n), layout, int(brute), int(threads), C.byref(bo))
    assert rc == 0
    return out


def flags_only(model, boxes, params, soa, layout=A.LAYOUT_POSE7, threads=1):
    """cpu_baseline leg: the predicate alone (flags), as isStateValid returns it."""
    from motion_planning_amd.engine import boxes_from_array
    lib = load()
    soa = np.ascontiguousarray(soa, dtype=np.float64)
    rows, n = soa.shape
    barr, m = boxes_from_array(boxes)
    flags = np.zeros(n, np.uint8)
    bo = BatchOut()
    bo.flags = flags.ctypes.data
    rc = lib.orc_eval_batch(C.byref(model), barr, C.c_size_t(m), C.byref(params), _p(soa), C.c_size_t(n),
                            C.c_size_t(n), layout, 0, int(threads), C.byref(bo))
    assert rc == 0
    return flags


def check_motion_batch(model, boxes, params, s1, s2, nd):
    from motion_planning_amd.engine import boxes_from_array
    lib = load()
    s1 = np.ascontiguousarray(s1, dtype=np.float64)
    s2 = np.ascontiguousarray(s2, dtype=np.float64)
    rows, E = s1.shape
    barr, m = boxes_from_array(boxes)
    ok = np.zeros(E, np.uint8)
    first = np.zeros(E, np.int32)
    band = np.zeros(E, np.uint8)
    if np.isscalar(nd):
        nd_arr, nd_u = None, int(nd)
    else:
        nd_arr, nd_u = np.ascontiguousarray(nd, dtype=np.int32), 0
    lib.orc_check_motion_batch(C.byref(model), barr, C.c_size_t(m), C.byref(params), _p(s1), _p(s2), C.c_size_t(E),
                               C.c_size_t(E), _p(nd_arr), nd_u, _p(ok), _p(first), _p(band))
    return ok, first, band


class Space:
    def __init__(self, params, J):
        self.lib, self.p, self.J = load(), params, J
        self.dim = self.lib.orc_space_dim(C.byref(params), J)

    def distance(self, a, b):
        a, b = np.ascontiguousarray(a, float), np.ascontiguousarray(b, float)
        return self.lib.orc_space_distance(C.byref(self.p), self.J, _p(a), _p(b))

    def valid_segment_count(self, a, b):
        a, b = np.ascontiguousarray(a, float), np.ascontiguousarray(b, float)
        return self.lib.orc_space_valid_segment_count(C.byref(self.p), self.J, _p(a), _p(b))

    def interpolate(self, a, b, t):
        a, b = np.ascontiguousarray(a, float), np.ascontiguousarray(b, float)
        out = np.empty(self.dim)
        self.lib.orc_space_interpolate(C.byref(self.p), self.J, _p(a), _p(b), C.c_double(t), _p(out))
        return out

    def max_extent(self):
        return self.lib.orc_space_max_extent(C.byref(self.p), self.J)

    def unpack(self, s):
        s = np.ascontiguousarray(s, float)
        out = np.empty(7 + self.J)
        self.lib.orc_unpack(C.byref(self.p), self.J, _p(s), _p(out))
        return out


def prim_poses(model, params, pose7j):
    lib = load()
    st = np.ascontiguousarray(pose7j, float)
    c = np.zeros((model.n_prim, 3))
    R = np.zeros((model.n_prim, 9))
    lib.orc_prim_poses(C.byref(model), C.byref(params), _p(st), _p(c), _p(R))
    return c, R.reshape(-1, 3, 3)


def prim_box_signed_dist(ptype, pc, pR, dim, box_row):
    from motion_planning_amd.engine import boxes_from_array
    lib = load()
    barr, _ = boxes_from_array(np.asarray(box_row, float).reshape(1, 10))
    pc = np.ascontiguousarray(pc, float); pR = np.ascontiguousarray(pR, float).reshape(9); dim = np.ascontiguousarray(dim, float)
    return lib.orc_prim_box_signed_dist(int(ptype), _p(pc), _p(pR), _p(dim), barr)


def scene_horizontal_gap(gap_left_x=0.0, gap_y_offset=0.6, gap_left_width=0.3, gap_right_width=0.3):
    from motion_planning_amd.engine import boxes_to_array
    lib = load()
    out = (A.Box * 4)()
    n = lib.orc_scene_horizontal_gap(C.c_double(gap_left_x), C.c_double(gap_y_offset), C.c_double(gap_left_width),
                                     C.c_double(gap_right_width), out)
    return boxes_to_array(out, n)


def scene_vertical_gap(gx=1.0, gy=1.0, gh=0.3, planning_walls=True, ceiling_offset=0.6, side_wall_width=2.0):
    from motion_planning_amd.engine import boxes_to_array
    lib = load()
    out = (A.Box * 10)()
    n = lib.orc_scene_vertical_gap(C.c_double(gx), C.c_double(gy), C.c_double(gh), int(planning_walls),
                                   C.c_double(ceiling_offset), C.c_double(side_wall_width), out)
    return boxes_to_array(out, n)


def avoid_batch(model, boxes, params, soa, damping_gain=0.01, constraint_range=0.03, forbidden_range=0.01, full_body=True):
    """CollisionAvoidance::getConstraint restated (collision_avoidance.cpp:102-277); soa = unpacked POSE7 states."""
    from motion_planning_amd.engine import boxes_from_array
    lib = load()
    soa = np.ascontiguousarray(soa, dtype=np.float64)
    rows, n = soa.shape
    assert rows == 7 + model.n_joint
    P, W = model.n_prim, 6 + model.n_joint
    barr, m = boxes_from_array(boxes)
    out = dict(dist=np.zeros((P, n)), box=np.zeros((P, n), np.int32), p_env=np.zeros((3 * P, n)),
               p_robot=np.zeros((3 * P, n)), row=np.zeros((W * P, n)), lb=np.zeros((P, n)), ok=np.zeros(n, np.uint8))
    lib.orc_avoid_batch(C.byref(model), barr, C.c_size_t(m), C.byref(params), _p(soa), C.c_size_t(n), C.c_size_t(n),
                        C.c_double(damping_gain), C.c_double(constraint_range), C.c_double(forbidden_range), int(full_body),
                        _p(out["dist"]), _p(out["box"]), _p(out["p_env"]), _p(out["p_robot"]), _p(out["row"]),
                        _p(out["lb"]), _p(out["ok"]))
    return out
