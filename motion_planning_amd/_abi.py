"""ctypes mirror of include/mvk.h (structs, enums, prototypes).

Plumbing only: no arithmetic happens in Python.  The struct layouts are checked
against the library's own sizeof() at load time.
"""
import ctypes as C
import os

MAX_SEG, MAX_ROTOR, MAX_JOINT, MAX_PRIM = 16, 8, 14, 32
MAX_STATE_DIM = 7 + MAX_JOINT

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_IO, ERR_NO_SOLUTION = 0, -1, -2, -3, -4, -5, -6
MODE_ONLY_JOINTS, MODE_ONLY_BASE, MODE_JOINTS_AND_BASE = 0, 1, 2
SE2, SE3 = 0, 1
ROBOT_HYDRUS, ROBOT_DRAGON = 0, 1
PRIM_SPHERE, PRIM_CAPSULE, PRIM_BOX = 0, 1, 2
FRAME_SEG, FRAME_GIMBAL_ROLL, FRAME_GIMBAL_PITCH = 0, 1, 2
LAYOUT_POSE7, LAYOUT_OMPL = 0, 1
NOALLOC_MAX_STATES = 65536
PRECISION_F64, PRECISION_F32_REFINED = 0, 1

d, i32 = C.c_double, C.c_int32


class Model(C.Structure):
    _fields_ = [
        ("abi_version", i32), ("robot_type", i32),
        ("n_seg", i32), ("n_joint", i32), ("n_rotor", i32), ("n_prim", i32),
        ("wrench_dof", i32), ("has_gimbal", i32),
        ("seg_axis", i32 * MAX_SEG), ("seg_joint", i32 * MAX_SEG), ("seg_link", i32 * MAX_SEG),
        ("seg_t", (d * 3) * MAX_SEG), ("seg_mass", d * MAX_SEG), ("seg_com", (d * 3) * MAX_SEG),
        ("rotor_seg", i32 * MAX_ROTOR), ("rotor_pos", (d * 3) * MAX_ROTOR), ("rotor_sigma", d * MAX_ROTOR), ("rotor_roll_z", d),
        ("edf_offset_y", d), ("edf_radius", d), ("edf_max_tilt", d),
        ("gravity", d), ("m_f_rate", d), ("thrust_min", d), ("thrust_max", d),
        ("fc_t_min_thre", d), ("fc_rp_min_thre", d), ("link_length", d),
        ("prim_type", i32 * MAX_PRIM), ("prim_seg", i32 * MAX_PRIM), ("prim_frame", i32 * MAX_PRIM),
        ("prim_rotor", i32 * MAX_PRIM),
        ("prim_pos", (d * 3) * MAX_PRIM), ("prim_rot", (d * 9) * MAX_PRIM), ("prim_dim", (d * 3) * MAX_PRIM),
        ("joint_low", d * MAX_JOINT), ("joint_high", d * MAX_JOINT),
    ]


class Box(C.Structure):
    _fields_ = [("center", d * 3), ("quat", d * 4), ("size", d * 3)]


class Params(C.Structure):
    _fields_ = [
        ("planning_mode", i32), ("motion_type", i32), ("check_stability", i32), ("check_collision", i32),
        ("eps_dist", d), ("eps_margin", d),
        ("pos_low", d * 3), ("pos_high", d * 3),
        ("joint_low", d * MAX_JOINT), ("joint_high", d * MAX_JOINT),
        ("validity_resolution", d), ("valid_segment_count_factor", i32), ("mode2_ompl_defaults", i32),
        ("joint_subspace_weight", d),
    ]


class Result(C.Structure):
    _fields_ = [("valid", C.c_uint8), ("band", C.c_uint8), ("stable", C.c_uint8), ("collision", C.c_uint8),
                ("min_dist", d), ("margin", d)]


class DetailOut(C.Structure):
    _fields_ = [("fc_t_min", C.c_void_p), ("fc_rp_min", C.c_void_p), ("thrust", C.c_void_p),
                ("rp_position_margin", C.c_void_p), ("cog", C.c_void_p), ("gimbal", C.c_void_p)]


class AvoidParams(C.Structure):
    _fields_ = [("damping_gain", d), ("constraint_range", d), ("forbidden_range", d), ("full_body", i32)]


class AvoidOut(C.Structure):
    _fields_ = [(k, C.c_void_p) for k in ("dist", "box", "p_env", "p_robot", "row", "lb", "ok")]


class Stats(C.Structure):
    _fields_ = [("n", C.c_uint64), ("valid_count", C.c_uint64), ("band_count", C.c_uint64),
                ("min_dist", d), ("min_dist_index", C.c_int64),
                ("min_margin", d), ("min_margin_index", C.c_int64),
                ("max_thrust", d), ("max_thrust_index", C.c_int64),
                ("min_thrust", d), ("min_thrust_index", C.c_int64),
                ("min_rp_position_margin", d), ("min_rp_position_margin_index", C.c_int64)]


CHECK_STATES_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.c_size_t, C.c_size_t, C.POINTER(C.c_uint8))
CHECK_MOTIONS_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_double), C.c_size_t, C.c_size_t,
                               C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_int32))


class Checker(C.Structure):
    _fields_ = [("check_states", CHECK_STATES_FN), ("check_motions", CHECK_MOTIONS_FN), ("user", C.c_void_p),
                ("check_motions_begin", C.c_void_p), ("check_motions_end", C.c_void_p)]   # optional split interface (NULL here)


class PlanRequest(C.Structure):
    _fields_ = [("start", C.POINTER(C.c_double)), ("goal", C.POINTER(C.c_double)), ("range", d), ("time_limit_s", d),
                ("seed", C.c_uint64), ("max_iterations", i32), ("batch", i32)]


class PlanResult(C.Structure):
    _fields_ = [("solved", i32), ("n_waypoints", i32), ("solve_ms", d), ("path_length", d),
                ("iterations", C.c_uint64), ("state_checks", C.c_uint64), ("edge_checks", C.c_uint64),
                ("edge_steps", C.c_uint64), ("tree_sizes", i32 * 2)]


class PathLog(C.Structure):
    _fields_ = [("n_states", i32), ("n_joint", i32), ("planning_mode", i32),
                ("planning_time", d), ("motion_cost", d), ("minimum_var", d), ("minimum_var_state_entry", i32),
                ("start", d * MAX_STATE_DIM), ("goal", d * MAX_STATE_DIM)]


vp, sz, ip = C.c_void_p, C.c_size_t, C.c_int
PM, PP, PB = C.POINTER(Model), C.POINTER(Params), C.POINTER(Box)

# every symbol include/mvk.h declares: name -> (restype, argtypes)
PROTOTYPES = {
    "mvk_abi_version": (ip, []),
    "mvk_model_sizeof": (sz, []),
    "mvk_params_sizeof": (sz, []),
    "mvk_last_error": (C.c_char_p, []),
    "mvk_device_count": (ip, []),
    "mvk_model_hydrus": (ip, [ip, PM]),
    "mvk_model_dragon": (ip, [ip, PM]),
    "mvk_params_default": (ip, [PM, ip, ip, PP]),
    "mvk_scene_horizontal_gap": (ip, [d, d, d, d, PB, ip]),
    "mvk_scene_vertical_gap": (ip, [d, d, d, ip, d, d, PB, ip]),
    "mvk_broadphase_query": (ip, [PB, sz, d, vp, sz, vp, vp, ip]),
    "mvk_far_grid_query": (ip, [PB, sz, d, d, vp, sz, vp, vp, ip]),
    "mvk_create": (vp, [PM, PB, sz, PP, ip]),
    "mvk_destroy": (None, [vp]),
    "mvk_set_scene": (ip, [vp, PB, sz]),
    "mvk_set_params": (ip, [vp, PP]),
    "mvk_get_params": (ip, [vp, PP]),
    "mvk_set_flag_precision": (ip, [vp, ip]),
    "mvk_last_refined_count": (ip, [vp, C.POINTER(C.c_uint64)]),
    "mvk_set_profiling": (ip, [vp, ip]),
    "mvk_last_kernel_ms": (ip, [vp, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "mvk_state_dim": (ip, [vp, ip]),
    "mvk_is_valid": (ip, [vp, vp, C.POINTER(Result)]),
    "mvk_is_valid_batch": (ip, [vp, vp, sz, sz, ip, vp, vp, vp, vp]),
    "mvk_is_valid_batch_dev": (ip, [vp, vp, sz, sz, ip, vp, vp, vp, vp, vp]),
    "mvk_is_valid_mask_dev": (ip, [vp, vp, sz, sz, ip, vp, vp]),
    "mvk_is_valid_mask_noalloc_dev": (ip, [vp, vp, sz, sz, ip, vp, vp]),
    "mvk_sweep_mask_dev": (ip, [vp, vp, C.c_uint64, sz, C.c_uint64, vp, vp]),
    "mvk_sweep_states": (ip, [PP, ip, vp, C.c_uint64, sz, C.c_uint64, vp, sz]),
    "mvk_mask_count_dev": (ip, [vp, sz, vp, vp]),
    "mvk_eval_batch": (ip, [vp, vp, sz, sz, ip, vp, vp, vp, vp, C.POINTER(DetailOut)]),
    "mvk_eval_batch_dev": (ip, [vp, vp, sz, sz, ip, vp, vp, vp, vp, C.POINTER(DetailOut), vp]),
    "mvk_eval_batch_f32_dev": (ip, [vp, vp, sz, sz, ip, vp, vp, vp, vp, vp, vp]),
    "mvk_f32_distance_tolerance": (d, [vp, d]),
    "mvk_reduce_stats": (ip, [vp, C.POINTER(Stats)]),
    "mvk_avoid_params_default": (ip, [C.POINTER(AvoidParams)]),
    "mvk_collision_avoidance_batch": (ip, [vp, vp, sz, sz, ip, C.POINTER(AvoidParams), C.POINTER(AvoidOut)]),
    "mvk_collision_avoidance_batch_dev": (ip, [vp, vp, sz, sz, ip, C.POINTER(AvoidParams), C.POINTER(AvoidOut), vp]),
    "mvk_check_motion_batch": (ip, [vp, vp, vp, sz, sz, vp, i32, vp, vp]),
    "mvk_check_motion_batch_begin": (ip, [vp, vp, vp, sz, sz, vp, i32]),
    "mvk_check_motion_batch_end": (ip, [vp, vp, vp]),
    "mvk_check_motion_batch_dev": (ip, [vp, vp, vp, sz, sz, vp, i32, vp, vp, vp]),
    "mvk_space_dim": (ip, [PP, ip]),
    "mvk_space_distance": (d, [PP, ip, vp, vp]),
    "mvk_space_interpolate": (ip, [PP, ip, vp, vp, d, vp]),
    "mvk_space_valid_segment_count": (ip, [PP, ip, vp, vp]),
    "mvk_space_max_extent": (d, [PP, ip]),
    "mvk_space_unpack": (ip, [PP, ip, vp, vp]),
    "mvk_plan_rrt_connect_cb": (ip, [PP, ip, C.POINTER(Checker), C.POINTER(PlanRequest), vp, ip, C.POINTER(PlanResult)]),
    "mvk_plan_rrt_connect": (ip, [vp, C.POINTER(PlanRequest), vp, ip, C.POINTER(PlanResult)]),
    "mvk_path_densify": (ip, [PP, ip, vp, ip, vp, ip]),
    "mvk_root_to_cog": (ip, [PM, vp, vp]),
    "mvk_cog_to_root": (ip, [PM, vp, vp]),
    "mvk_path_smooth": (ip, [vp, ip, ip, d, vp, ip]),
    "mvk_path_resample": (ip, [vp, ip, ip, d, d, vp, ip]),
    "mvk_spline_create": (vp, [vp, ip, ip, d, ip]),
    "mvk_spline_destroy": (None, [vp]),
    "mvk_spline_dim": (ip, [vp]),
    "mvk_spline_eval": (ip, [vp, vp, ip, ip, vp]),
    "mvk_spline_eval_dev": (ip, [vp, ip, vp, sz, ip, vp, sz, vp]),
    "mvk_path_log_save": (ip, [C.c_char_p, C.POINTER(PathLog), vp]),
    "mvk_path_log_load": (ip, [C.c_char_p, ip, C.POINTER(PathLog), vp, ip]),
}

_LIB = None


def lib_path():
    if os.environ.get("MVK_LIB"):   # diagnostic builds (profiles/diag.py); never set by tests or bench.py
        return os.environ["MVK_LIB"]
    return os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", "libmvk.so")


PROTOTYPES.update({
    "mvk_comm_unique_id": (ip, [vp]),
    "mvk_comm_create": (ip, [vp, ip, ip, ip, C.POINTER(vp)]),
    "mvk_comm_destroy": (ip, [vp]),
    "mvk_comm_rank": (ip, [vp]),
    "mvk_comm_world": (ip, [vp]),
    "mvk_shard_range": (ip, [sz, ip, ip, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]),
    "mvk_gather_mask_dev": (ip, [vp, vp, sz, vp, vp]),
    "mvk_allreduce_count_dev": (ip, [vp, vp, sz, vp]),
    "mvk_exchange_masks_dev": (ip, [vp, ip, vp, sz, vp, vp, vp, vp]),
    "mvk_exchange_wait": (ip, [vp, ip]),
    "mvk_allreduce_stats": (ip, [vp, C.POINTER(Stats), C.c_uint64, C.POINTER(Stats), vp]),
    "mvk_merge_stats": (ip, [C.POINTER(Stats), C.POINTER(C.c_uint64), ip, C.POINTER(Stats)]),
    # pipelined batches (csrc/mvk_pipe.cpp)
    "mvk_pipe_create": (ip, [PM, PB, sz, PP, ip, ip, C.POINTER(vp)]),
    "mvk_pipe_destroy": (None, [vp]),
    "mvk_pipe_depth": (ip, [vp]),
    "mvk_pipe_ctx": (vp, [vp, ip]),
    "mvk_pipe_stream": (vp, [vp, ip]),
    "mvk_pipe_set_scene": (ip, [vp, PB, sz]),
    "mvk_pipe_is_valid_mask": (ip, [vp, vp, sz, sz, ip, vp, vp, C.POINTER(ip)]),
    "mvk_pipe_sweep_mask": (ip, [vp, vp, C.c_uint64, sz, C.c_uint64, vp, C.POINTER(ip)]),
    "mvk_pipe_eval_batch": (ip, [vp, vp, sz, sz, ip, vp, vp, vp, vp, vp, C.POINTER(ip)]),
    "mvk_pipe_wait": (ip, [vp, ip, vp]),
    "mvk_pipe_drain": (ip, [vp]),
})
PIPE_NO_PRODUCER = C.c_void_p(-1 & (2 ** (8 * C.sizeof(C.c_void_p)) - 1))   # MVK_PIPE_NO_PRODUCER / MVK_PIPE_HOST
COMM_ID_BYTES = 128

# every symbol include/mvk_debug.h declares (diagnostics, not part of the drop-in boundary)
DEBUG_PROTOTYPES = {
    "mvk_debug_f32_bounds": (ip, [vp, vp, sz, sz, ip, vp, vp, vp, vp, vp]),
    "mvk_debug_fused_giveups": (C.c_longlong, [vp, ip]),
    "mvk_debug_grid_bounds": (ip, [PB, sz, d, d, vp, sz, vp, vp, vp, ip]),
    "mvk_debug_pipe_restreamed": (ip, [vp]),
    "mvk_debug_pipe_lanes_overlap": (ip, [vp, ip, ip]),
}


def load():
    """Load libmvk.so (built in-tree by __graft_entry__.build()).  Fails loudly."""
    global _LIB
    if _LIB is not None:
        return _LIB
    path = lib_path()
    if not os.path.exists(path):
        raise RuntimeError(
            f"{path} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(there is no Python / CPU fallback for the validity kernels)")
    # A process must hold ONE HIP runtime.  PyTorch-ROCm bundles its own libamdhip64; when tensors are going
    # to be shared with this library (tests, bench.py) torch has to be loaded first so that libmvk.so binds to
    # the runtime torch already brought in (loading /opt/rocm's copy first leaves torch without devices).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, (res, args) in list(PROTOTYPES.items()) + list(DEBUG_PROTOTYPES.items()):
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    if lib.mvk_abi_version() != 1:
        raise RuntimeError("libmvk.so ABI version mismatch")
    if lib.mvk_model_sizeof() != C.sizeof(Model) or lib.mvk_params_sizeof() != C.sizeof(Params):
        raise RuntimeError("ctypes struct layout differs from the C ABI")
    _LIB = lib
    return lib


_HIP = None


def hip_runtime():
    """libamdhip64 through ctypes (tests: raw streams outside torch's pool)"""
    global _HIP
    if _HIP is None:
        for name in ("libamdhip64.so", "/opt/rocm/lib/libamdhip64.so"):
            try:
                _HIP = C.CDLL(name)
                break
            except OSError:
                continue
        if _HIP is None:
            raise RuntimeError("libamdhip64.so not found")
        _HIP.hipStreamCreate.argtypes = [C.POINTER(C.c_void_p)]
        _HIP.hipStreamSynchronize.argtypes = [C.c_void_p]
        _HIP.hipStreamDestroy.argtypes = [C.c_void_p]
    return _HIP
