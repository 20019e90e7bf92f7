"""Thin Python handle over the C ABI (include/mvk.h) for tests and bench.py.

All arithmetic runs in libmvk.so on the GPU; this module only marshals numpy
arrays / torch tensors into the plain-pointer calls.  There is no fallback: if
the library or a gfx950 device is missing the calls raise.
"""
import ctypes as C

import numpy as np

from . import _abi as A


class MvkError(RuntimeError):
    pass


def _check(lib, rc):
    if rc != 0:
        raise MvkError(f"mvk error {rc}: {lib.mvk_last_error().decode()}")


def make_model(kind, n_links):
    lib = A.load()
    m = A.Model()
    fn = {"hydrus": lib.mvk_model_hydrus, "dragon": lib.mvk_model_dragon}[kind]
    _check(lib, fn(n_links, C.byref(m)))
    return m


def default_params(model, planning_mode, motion_type):
    lib = A.load()
    p = A.Params()
    _check(lib, lib.mvk_params_default(C.byref(model), planning_mode, motion_type, C.byref(p)))
    return p


def boxes_from_array(arr):
    """arr: [M, 10] = centre xyz, quaternion xyzw, full side lengths."""
    arr = np.ascontiguousarray(arr, dtype=np.float64).reshape(-1, 10)
    out = (A.Box * max(len(arr), 1))()
    for k, row in enumerate(arr):
        out[k].center[:] = row[0:3]
        out[k].quat[:] = row[3:7]
        out[k].size[:] = row[7:10]
    return out, len(arr)


def boxes_to_array(boxes, n):
    return np.array([[*boxes[k].center, *boxes[k].quat, *boxes[k].size] for k in range(n)], dtype=np.float64).reshape(n, 10)


def horizontal_gap_scene(gap_left_x=0.0, gap_y_offset=0.6, gap_left_width=0.3, gap_right_width=0.3):
    """SamplingBasedMethod::setCollisionWorld defaults (sampling_based_method.cpp:84-87)."""
    lib = A.load()
    out = (A.Box * 4)()
    n = lib.mvk_scene_horizontal_gap(gap_left_x, gap_y_offset, gap_left_width, gap_right_width, out, 4)
    if n < 0:
        _check(lib, n)
    return boxes_to_array(out, n)


def vertical_gap_scene(gap_x_width=1.0, gap_y_width=1.0, gap_height=0.3, planning_walls=True,
                       ceiling_offset=0.6, side_wall_width=2.0):
    lib = A.load()
    out = (A.Box * 10)()
    n = lib.mvk_scene_vertical_gap(gap_x_width, gap_y_width, gap_height, int(planning_walls), ceiling_offset,
                                   side_wall_width, out, 10)
    if n < 0:
        _check(lib, n)
    return boxes_to_array(out, n)


def broadphase_query(boxes, radius, points, cap=256):
    """Host diagnostic: ids the broadphase grid lists for each query point ([n, 3])."""
    lib = A.load()
    arr, m = boxes_from_array(boxes)
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    counts = np.zeros(len(pts), np.int32)
    ids = np.full((len(pts), cap), -1, np.int32)
    _check(lib, lib.mvk_broadphase_query(arr, m, float(radius), pts.ctypes.data_as(C.c_void_p), len(pts),
                                         counts.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), cap))
    return counts, ids


def grid_bounds_query(boxes, radius, pad, points, cap=1024):
    """Host diagnostic (include/mvk_debug.h): a distance grid's list for each query point in list order, and per entry
    the lower bound its group carries."""
    lib = A.load()
    arr, m = boxes_from_array(boxes)
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    counts = np.zeros(len(pts), np.int32)
    ids = np.full((len(pts), cap), -1, np.int32)
    bounds = np.full((len(pts), cap), np.nan, np.float32)
    _check(lib, lib.mvk_debug_grid_bounds(arr, m, float(radius), float(pad), pts.ctypes.data_as(C.c_void_p), len(pts),
                                          counts.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p),
                                          bounds.ctypes.data_as(C.c_void_p), cap))
    return counts, ids, bounds


def far_grid_query(boxes, reach, margin, points, cap=1024):
    """Host diagnostic: ids the far grid lists for each query point ([n, 3]); counts -1 outside its volume."""
    lib = A.load()
    arr, m = boxes_from_array(boxes)
    pts = np.ascontiguousarray(points, dtype=np.float64).reshape(-1, 3)
    counts = np.zeros(len(pts), np.int32)
    ids = np.full((len(pts), cap), -1, np.int32)
    _check(lib, lib.mvk_far_grid_query(arr, m, float(reach), float(margin), pts.ctypes.data_as(C.c_void_p), len(pts),
                                       counts.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p), cap))
    return counts, ids


def sweep_states(params, n_joint, root_pose7, first_index, n, seed):
    """The states of the C-space sweep, materialised on the host ([7+J, n], POSE7 rows); no device needed."""
    lib = A.load()
    pose = np.ascontiguousarray(root_pose7, np.float64)
    out = np.zeros((7 + n_joint, n))
    _check(lib, lib.mvk_sweep_states(C.byref(params), n_joint, pose.ctypes.data_as(C.c_void_p), int(first_index), int(n),
                                     int(seed), out.ctypes.data_as(C.c_void_p), n))
    return out


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class Space:
    """OMPL state-space arithmetic of (params, J) -- host only, no device."""

    def __init__(self, params, n_joint):
        self.lib = A.load()
        self.p = params
        self.J = n_joint
        self.dim = self.lib.mvk_space_dim(C.byref(params), n_joint)

    def _v(self, s):
        s = np.ascontiguousarray(s, dtype=np.float64)
        assert s.shape == (self.dim,), (s.shape, self.dim)
        return s

    def distance(self, a, b):
        a, b = self._v(a), self._v(b)
        return self.lib.mvk_space_distance(C.byref(self.p), self.J, _ptr(a), _ptr(b))

    def valid_segment_count(self, a, b):
        a, b = self._v(a), self._v(b)
        return self.lib.mvk_space_valid_segment_count(C.byref(self.p), self.J, _ptr(a), _ptr(b))

    def interpolate(self, a, b, t):
        a, b = self._v(a), self._v(b)
        out = np.empty(self.dim)
        _check(self.lib, self.lib.mvk_space_interpolate(C.byref(self.p), self.J, _ptr(a), _ptr(b), float(t), _ptr(out)))
        return out

    def max_extent(self):
        return self.lib.mvk_space_max_extent(C.byref(self.p), self.J)

    def unpack(self, s):
        s = self._v(s)
        out = np.empty(7 + self.J)
        _check(self.lib, self.lib.mvk_space_unpack(C.byref(self.p), self.J, _ptr(s), _ptr(out)))
        return out


class Pipe:
    """mvk_pipe (include/mvk.h, pipelined batches): `depth` contexts with a stream each, consecutive batches handed to
    them round-robin so that one batch's exact stage runs beside the next batch's screen stage.  ctypes plumbing."""

    def __init__(self, model, boxes, params, depth=2, device=0):
        self.lib = A.load()
        self.model, self.params = model, params
        arr, n = boxes_from_array(boxes if boxes is not None else np.zeros((0, 10)))
        self.pipe = C.c_void_p()
        _check(self.lib, self.lib.mvk_pipe_create(C.byref(model), arr, n, C.byref(params), device, int(depth), C.byref(self.pipe)))
        self.depth = self.lib.mvk_pipe_depth(self.pipe)

    def close(self):
        if getattr(self, "pipe", None):
            self.lib.mvk_pipe_destroy(self.pipe)
            self.pipe = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _producer(stream):
        return A.PIPE_NO_PRODUCER if stream is None else C.c_void_p(stream.cuda_stream)

    def set_scene(self, boxes):
        arr, n = boxes_from_array(boxes)
        _check(self.lib, self.lib.mvk_pipe_set_scene(self.pipe, arr, n))

    def is_valid_mask(self, d_soa, layout, d_mask, producer_stream=None):
        """submit one batch; returns its ticket.  producer_stream: torch stream the inputs are produced on (None: ready)"""
        rows, n = d_soa.shape
        t = C.c_int(-1)
        _check(self.lib, self.lib.mvk_pipe_is_valid_mask(self.pipe, C.c_void_p(d_soa.data_ptr()), n, d_soa.stride(0), layout,
                                                         C.c_void_p(d_mask.data_ptr()), self._producer(producer_stream), C.byref(t)))
        return t.value

    def sweep_mask(self, root_pose7, first_index, n, seed, d_mask):
        pose = np.ascontiguousarray(root_pose7, np.float64)
        t = C.c_int(-1)
        _check(self.lib, self.lib.mvk_pipe_sweep_mask(self.pipe, _ptr(pose), int(first_index), int(n), int(seed),
                                                      C.c_void_p(d_mask.data_ptr()), C.byref(t)))
        return t.value

    def eval_batch(self, d_soa, layout, d_flags=None, d_min_dist=None, d_margin=None, d_band=None, producer_stream=None):
        rows, n = d_soa.shape
        dp = lambda x: C.c_void_p(x.data_ptr()) if x is not None else None   # noqa: E731
        t = C.c_int(-1)
        _check(self.lib, self.lib.mvk_pipe_eval_batch(self.pipe, C.c_void_p(d_soa.data_ptr()), n, d_soa.stride(0), layout, dp(d_flags),
                                                      dp(d_min_dist), dp(d_margin), dp(d_band), self._producer(producer_stream), C.byref(t)))
        return t.value

    def wait(self, ticket, consumer_stream=None):
        """consumer_stream (torch stream): ordered behind the ticket's lane on the GPU; None: block the host"""
        _check(self.lib, self.lib.mvk_pipe_wait(self.pipe, int(ticket),
                                                A.PIPE_NO_PRODUCER if consumer_stream is None else C.c_void_p(consumer_stream.cuda_stream)))

    def drain(self):
        _check(self.lib, self.lib.mvk_pipe_drain(self.pipe))


class Engine:
    """Owns one mvk_ctx (one GPU).  Mirrors the role of sampling_base::MotionPlanning's
    robot_model_ptr_ + planning_scene_ for the validity path."""

    def __init__(self, model, boxes, params, device=0):
        self.lib = A.load()
        self.model, self.params = model, params
        arr, n = boxes_from_array(boxes if boxes is not None else np.zeros((0, 10)))
        self._boxes = arr
        self.ctx = self.lib.mvk_create(C.byref(model), arr, n, C.byref(params), device)
        if not self.ctx:
            raise MvkError(f"mvk_create failed: {self.lib.mvk_last_error().decode()}")
        self.n_joint = model.n_joint
        self.n_rotor = model.n_rotor

    def close(self):
        if getattr(self, "ctx", None):
            self.lib.mvk_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_scene(self, boxes):
        arr, n = boxes_from_array(boxes)
        _check(self.lib, self.lib.mvk_set_scene(self.ctx, arr, n))

    def set_params(self, params):
        _check(self.lib, self.lib.mvk_set_params(self.ctx, C.byref(params)))
        self.params = params

    def set_flag_precision(self, precision):
        """A.PRECISION_F64 (default) or A.PRECISION_F32_REFINED: arithmetic of the flag path (include/mvk.h)"""
        _check(self.lib, self.lib.mvk_set_flag_precision(self.ctx, int(precision)))

    def last_refined_count(self):
        v = C.c_uint64()
        _check(self.lib, self.lib.mvk_last_refined_count(self.ctx, C.byref(v)))
        return int(v.value)

    def set_profiling(self, enable=True):
        _check(self.lib, self.lib.mvk_set_profiling(self.ctx, int(enable)))

    def last_kernel_ms(self):
        """(first kernel ms, second kernel ms) of the last batched call; needs set_profiling(True)."""
        a, b = C.c_float(), C.c_float()
        _check(self.lib, self.lib.mvk_last_kernel_ms(self.ctx, C.byref(a), C.byref(b)))
        return a.value, b.value

    def state_dim(self, layout):
        return self.lib.mvk_state_dim(self.ctx, layout)

    # ---- host-pointer entry points ------------------------------------------
    def is_valid(self, state):
        s = np.ascontiguousarray(state, dtype=np.float64)
        r = A.Result()
        _check(self.lib, self.lib.mvk_is_valid(self.ctx, _ptr(s), C.byref(r)))
        return r

    def is_valid_batch(self, soa, layout=A.LAYOUT_POSE7, want=("flags",)):
        """soa: [rows, n] float64 (C order).  want: subset of flags,min_dist,margin,band."""
        soa = np.ascontiguousarray(soa, dtype=np.float64)
        rows, n = soa.shape
        assert rows == self.state_dim(layout), (rows, self.state_dim(layout))
        out = {}
        if "flags" in want: out["flags"] = np.zeros(n, np.uint8)
        if "min_dist" in want: out["min_dist"] = np.zeros(n)
        if "margin" in want: out["margin"] = np.zeros(n)
        if "band" in want: out["band"] = np.zeros(n, np.uint8)
        _check(self.lib, self.lib.mvk_is_valid_batch(self.ctx, _ptr(soa), n, n, layout, _ptr(out.get("flags")),
                                                     _ptr(out.get("min_dist")), _ptr(out.get("margin")),
                                                     _ptr(out.get("band"))))
        return out

    def eval_batch(self, soa, layout=A.LAYOUT_POSE7):
        soa = np.ascontiguousarray(soa, dtype=np.float64)
        rows, n = soa.shape
        assert rows == self.state_dim(layout)
        N = self.n_rotor
        out = dict(flags=np.zeros(n, np.uint8), min_dist=np.zeros(n), margin=np.zeros(n), band=np.zeros(n, np.uint8),
                   fc_t_min=np.zeros(n), fc_rp_min=np.zeros(n), thrust=np.zeros((N, n)),
                   rp_position_margin=np.zeros(n), cog=np.zeros((3, n)))
        if self.model.has_gimbal:
            out["gimbal"] = np.zeros((2 * N, n))
        det = A.DetailOut(_ptr(out["fc_t_min"]), _ptr(out["fc_rp_min"]), _ptr(out["thrust"]),
                          _ptr(out["rp_position_margin"]), _ptr(out["cog"]), _ptr(out.get("gimbal")))
        _check(self.lib, self.lib.mvk_eval_batch(self.ctx, _ptr(soa), n, n, layout, _ptr(out["flags"]),
                                                 _ptr(out["min_dist"]), _ptr(out["margin"]), _ptr(out["band"]),
                                                 C.byref(det)))
        return out

    def collision_avoidance_batch(self, soa, layout=A.LAYOUT_POSE7, damping_gain=0.01, constraint_range=0.03,
                                  forbidden_range=0.01, full_body=True):
        """CollisionAvoidance::getConstraint rows for a batch (include/mvk.h)."""
        soa = np.ascontiguousarray(soa, dtype=np.float64)
        rows, n = soa.shape
        assert rows == self.state_dim(layout)
        P, W = self.model.n_prim, 6 + self.n_joint
        out = dict(dist=np.zeros((P, n)), box=np.zeros((P, n), np.int32), p_env=np.zeros((3 * P, n)),
                   p_robot=np.zeros((3 * P, n)), row=np.zeros((W * P, n)), lb=np.zeros((P, n)), ok=np.zeros(n, np.uint8))
        prm = A.AvoidParams(damping_gain, constraint_range, forbidden_range, int(full_body))
        o = A.AvoidOut(*[_ptr(out[k]) for k in ("dist", "box", "p_env", "p_robot", "row", "lb", "ok")])
        _check(self.lib, self.lib.mvk_collision_avoidance_batch(self.ctx, _ptr(soa), n, n, layout, C.byref(prm), C.byref(o)))
        return out

    def reduce_stats(self):
        s = A.Stats()
        _check(self.lib, self.lib.mvk_reduce_stats(self.ctx, C.byref(s)))
        return {k: getattr(s, k) for k, _ in A.Stats._fields_}

    def check_motion_batch(self, s1, s2, nd):
        """s1, s2: [dim, E] OMPL-layout SoA; nd: int or int32 array [E]."""
        s1 = np.ascontiguousarray(s1, dtype=np.float64)
        s2 = np.ascontiguousarray(s2, dtype=np.float64)
        rows, E = s1.shape
        assert s2.shape == s1.shape and rows == self.state_dim(A.LAYOUT_OMPL)
        ok = np.zeros(E, np.uint8)
        first = np.zeros(E, np.int32)
        if np.isscalar(nd):
            nd_arr, nd_u = None, int(nd)
        else:
            nd_arr, nd_u = np.ascontiguousarray(nd, dtype=np.int32), 0
        _check(self.lib, self.lib.mvk_check_motion_batch(self.ctx, _ptr(s1), _ptr(s2), E, E, _ptr(nd_arr), nd_u,
                                                         _ptr(ok), _ptr(first)))
        return ok, first

    def check_motion_batch_begin(self, s1, s2, nd):
        """first half of check_motion_batch: returns once the transfers and kernels are enqueued"""
        s1 = np.ascontiguousarray(s1, dtype=np.float64)
        s2 = np.ascontiguousarray(s2, dtype=np.float64)
        rows, E = s1.shape
        assert s2.shape == s1.shape and rows == self.state_dim(A.LAYOUT_OMPL)
        if np.isscalar(nd):
            nd_arr, nd_u = None, int(nd)
        else:
            nd_arr, nd_u = np.ascontiguousarray(nd, dtype=np.int32), 0
        self._pending_edges = E
        _check(self.lib, self.lib.mvk_check_motion_batch_begin(self.ctx, _ptr(s1), _ptr(s2), E, E, _ptr(nd_arr), nd_u))

    def check_motion_batch_end(self):
        """second half: waits and returns (ok, first_invalid) of the last begin"""
        E = getattr(self, "_pending_edges", 0)
        ok = np.zeros(E, np.uint8)
        first = np.zeros(E, np.int32)
        _check(self.lib, self.lib.mvk_check_motion_batch_end(self.ctx, _ptr(ok), _ptr(first)))
        return ok, first

    # ---- device-pointer entry points (torch tensors resident in HBM) ----------
    @staticmethod
    def _dp(t):
        return None if t is None else C.c_void_p(t.data_ptr())

    @staticmethod
    def _stream(stream):
        if stream is None:
            import torch
            stream = torch.cuda.current_stream()
        return C.c_void_p(stream.cuda_stream)

    def is_valid_batch_dev(self, d_soa, layout, d_flags=None, d_min_dist=None, d_margin=None, d_band=None, stream=None):
        rows, n = d_soa.shape
        ld = d_soa.stride(0)
        assert d_soa.stride(1) == 1 and rows == self.state_dim(layout)
        _check(self.lib, self.lib.mvk_is_valid_batch_dev(self.ctx, self._dp(d_soa), n, ld, layout, self._dp(d_flags),
                                                         self._dp(d_min_dist), self._dp(d_margin), self._dp(d_band),
                                                         self._stream(stream)))

    def eval_batch_f32_dev(self, d_soa, layout, d_flags=None, d_min_dist=None, d_margin=None, d_margin_err=None, d_band=None,
                           stream=None):
        """the fp32 variant of the distance / margin outputs (mvk_eval_batch_f32_dev): float32 device tensors"""
        rows, n = d_soa.shape
        ld = d_soa.stride(0)
        assert d_soa.stride(1) == 1 and rows == self.state_dim(layout)
        _check(self.lib, self.lib.mvk_eval_batch_f32_dev(self.ctx, self._dp(d_soa), n, ld, layout, self._dp(d_flags),
                                                         self._dp(d_min_dist), self._dp(d_margin), self._dp(d_margin_err),
                                                         self._dp(d_band), self._stream(stream)))

    def f32_distance_tolerance(self, coord_abs_max):
        """the fp32 variant's distance tolerance for a state whose root position has that largest |coordinate|"""
        return float(self.lib.mvk_f32_distance_tolerance(self.ctx, float(coord_abs_max)))

    def collision_avoidance_batch_dev(self, d_soa, layout, d_out, damping_gain=0.01, constraint_range=0.03,
                                      forbidden_range=0.01, full_body=True, stream=None):
        """device-pointer form of collision_avoidance_batch: d_out maps any of dist [P, ld], box (int32) [P, ld],
        p_env / p_robot [3P, ld], row [(6+J)P, ld], lb [P, ld], ok (uint8) [ld] to device tensors whose row pitch
        is that of d_soa; outputs left out are not written"""
        rows, n = d_soa.shape
        ld = d_soa.stride(0)
        assert d_soa.stride(1) == 1 and rows == self.state_dim(layout)
        prm = A.AvoidParams(damping_gain, constraint_range, forbidden_range, int(full_body))
        o = A.AvoidOut(*[self._dp(d_out.get(k)) for k in ("dist", "box", "p_env", "p_robot", "row", "lb", "ok")])
        _check(self.lib, self.lib.mvk_collision_avoidance_batch_dev(self.ctx, self._dp(d_soa), n, ld, layout, C.byref(prm),
                                                                    C.byref(o), self._stream(stream)))

    def is_valid_mask_dev(self, d_soa, layout, d_mask, stream=None):
        rows, n = d_soa.shape
        ld = d_soa.stride(0)
        assert d_soa.stride(1) == 1 and rows == self.state_dim(layout)
        assert d_mask.numel() * 64 >= n
        _check(self.lib, self.lib.mvk_is_valid_mask_dev(self.ctx, self._dp(d_soa), n, ld, layout, self._dp(d_mask),
                                                        self._stream(stream)))

    def is_valid_mask_noalloc_dev(self, d_soa, layout, d_mask, stream=None):
        """mvk_is_valid_mask_dev for callers that cannot allocate (at most A.NOALLOC_MAX_STATES states per call)"""
        rows, n = d_soa.shape
        ld = d_soa.stride(0)
        assert d_soa.stride(1) == 1 and rows == self.state_dim(layout)
        assert d_mask.numel() * 64 >= n
        _check(self.lib, self.lib.mvk_is_valid_mask_noalloc_dev(self.ctx, self._dp(d_soa), n, ld, layout, self._dp(d_mask),
                                                                self._stream(stream)))

    def sweep_mask_dev(self, root_pose7, first_index, n, seed, d_mask, stream=None):
        """C-space sweep: states generated in the kernel from the sample index (include/mvk.h)."""
        pose = np.ascontiguousarray(root_pose7, np.float64)
        assert pose.shape == (7,) and d_mask.numel() * 64 >= n
        _check(self.lib, self.lib.mvk_sweep_mask_dev(self.ctx, _ptr(pose), int(first_index), int(n), int(seed),
                                                     self._dp(d_mask), self._stream(stream)))

    def sweep_states(self, root_pose7, first_index, n, seed):
        return sweep_states(self.params, self.n_joint, root_pose7, first_index, n, seed)

    def mask_count_dev(self, d_mask, n_bits, d_count, stream=None):
        _check(self.lib, self.lib.mvk_mask_count_dev(self._dp(d_mask), int(n_bits), self._dp(d_count), self._stream(stream)))

    def check_motion_batch_dev(self, d_s1, d_s2, nd, d_ok, d_first=None, stream=None):
        rows, E = d_s1.shape
        ld = d_s1.stride(0)
        assert d_s2.shape == d_s1.shape and d_s2.stride(0) == ld
        if isinstance(nd, int):
            d_nd, nd_u = None, nd
        else:
            d_nd, nd_u = nd, 0
        _check(self.lib, self.lib.mvk_check_motion_batch_dev(self.ctx, self._dp(d_s1), self._dp(d_s2), E, ld,
                                                             self._dp(d_nd), nd_u, self._dp(d_ok), self._dp(d_first),
                                                             self._stream(stream)))
