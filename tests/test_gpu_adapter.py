"""The OMPL-shaped C++ adapter (include/mvk_ompl_adapter.hpp) RUN on the GPU: what replaces
motion_planning.cpp:236-249.  motion_planning_amd/lib/libmvk_adapter_check.so (csrc/mvk_adapter_check.cpp, the
MVK_OMPL_STUB build of the header) drives the adapter through ompl::base base-class pointers on OMPL state objects;
its answers must equal the C ABI called through ctypes and the oracle."""
import ctypes as C
import os

import numpy as np
import pytest

import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A

pytestmark = pytest.mark.gpu


def check_lib():
    A.load()
    path = os.path.join(os.path.dirname(A.lib_path()), "libmvk_adapter_check.so")
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.mvk_adapter_check_run.argtypes = [vp, C.c_int, vp, vp, C.c_int] + [vp] * 9
    lib.mvk_adapter_check_errors.argtypes = [vp, C.c_int, vp, vp, vp]
    lib.mvk_adapter_check_clearance.argtypes = [vp, C.c_int, vp, C.c_int, vp, vp, vp, vp]
    return lib


def p_(a):
    return a.ctypes.data_as(C.c_void_p)


def run_adapter(eng, J, s1, s2):
    """s1, s2: [n, dim] row-major OMPL-layout states"""
    lib = check_lib()
    n, dim = s1.shape
    out = dict(valid_single=np.zeros(2 * n, np.uint8), valid_batch=np.zeros(2 * n, np.uint8), motion2=np.zeros(n, np.uint8),
               motion3=np.zeros(n, np.uint8), last_frac=np.zeros(n), last_state=np.zeros((n, dim)),
               motions_ok=np.zeros(n, np.uint8), motions_first=np.zeros(n, np.int32), counters=np.zeros(4, np.uint32))
    rc = lib.mvk_adapter_check_run(eng.ctx, J, p_(s1), p_(s2), n, *[p_(out[k]) for k in (
        "valid_single", "valid_batch", "motion2", "motion3", "last_frac", "last_state", "motions_ok", "motions_first", "counters")])
    assert rc == 0, rc
    return out


def cases():
    rng = np.random.default_rng(11)
    # c1: HYDRUS quad, mode 2 / SE2, the default horizontal gap world; short and long edges, some through the walls
    m = mp.make_model("hydrus", 4)
    p = mp.default_params(m, A.MODE_JOINTS_AND_BASE, A.SE2)
    n = 96
    s1 = np.column_stack([rng.uniform(-2, 3, n), rng.uniform(-1.5, 1.5, n), rng.uniform(-3.1, 3.1, n), rng.uniform(-1.57, 1.57, (n, 3))])
    s2 = s1 + np.column_stack([rng.normal(0, 0.8, n), rng.normal(0, 0.5, n), rng.normal(0, 0.4, n), rng.normal(0, 0.3, (n, 3))])
    s2[:, 2] = (s2[:, 2] + np.pi) % (2 * np.pi) - np.pi
    s2[:, 3:] = np.clip(s2[:, 3:], -1.58, 1.58)
    # a motion straight through the left wall (x = 0..0.3, y > 0.3) between two free states: must be invalid
    s1[0] = [-1.0, 1.2, 0.0, 1.57, 1.57, 1.57]
    s2[0] = [1.5, 1.2, 0.0, 1.57, 1.57, 1.57]
    yield "c1", m, p, mp.horizontal_gap_scene(), np.ascontiguousarray(s1), np.ascontiguousarray(s2)
    # DRAGON, mode 2 / SE3, the vertical gap world of the reference's logs (gap 0.5, height 0.7)
    m = mp.make_model("dragon", 4)
    p = mp.default_params(m, A.MODE_JOINTS_AND_BASE, A.SE3)
    p.pos_low[:] = [-1.0, -1.0, 0.0]
    p.pos_high[:] = [1.0, 1.0, 1.4]
    n = 64
    q = rng.normal(size=(n, 4)) * [0.15, 0.15, 0.5, 1.0]
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    joints = np.tile([0.0, 1.57, 0.0, 1.57, 0.0, 1.57], (n, 1)) + rng.normal(0, 0.15, (n, 6))
    s1 = np.column_stack([rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, n), rng.uniform(0.15, 1.2, n), q, joints])
    q2 = q + rng.normal(0, 0.1, (n, 4))
    q2 /= np.linalg.norm(q2, axis=1, keepdims=True)
    s2 = np.column_stack([s1[:, :3] + rng.normal(0, 0.2, (n, 3)), q2, np.clip(joints + rng.normal(0, 0.1, (n, 6)), -1.58, 1.58)])
    s2[:, 2] = np.clip(s2[:, 2], 0.05, 1.35)
    yield "dragon_se3", m, p, mp.vertical_gap_scene(0.5, 0.5, 0.7), np.ascontiguousarray(s1), np.ascontiguousarray(s2)
    # mode 1 (base only, SE2) and mode 0 (joints only): the other two state layouts of flatten / unflatten
    m = mp.make_model("hydrus", 4)
    p = mp.default_params(m, A.MODE_ONLY_BASE, A.SE2)
    p.validity_resolution = 0.005
    n = 32
    s1 = np.column_stack([rng.uniform(-2, 3, n), rng.uniform(-1.5, 1.5, n), rng.uniform(-3.1, 3.1, n)])
    s2 = s1 + rng.normal(0, 0.4, (n, 3))
    s2[:, 2] = (s2[:, 2] + np.pi) % (2 * np.pi) - np.pi
    yield "mode1", m, p, mp.horizontal_gap_scene(), np.ascontiguousarray(s1), np.ascontiguousarray(s2)
    p = mp.default_params(m, A.MODE_ONLY_JOINTS, A.SE2)
    s1 = rng.uniform(-1.57, 1.57, (n, 3))
    s2 = np.clip(s1 + rng.normal(0, 0.5, (n, 3)), -1.58, 1.58)
    yield "mode0", m, p, mp.horizontal_gap_scene(), np.ascontiguousarray(s1), np.ascontiguousarray(s2)


@pytest.mark.parametrize("case", list(cases()), ids=lambda c: c[0])
def test_adapter_equals_the_c_abi_and_the_oracle(case):
    name, m, p, boxes, s1, s2 = case
    J, n = m.n_joint, len(s1)
    eng = mp.Engine(m, boxes, p)
    got = run_adapter(eng, J, s1, s2)
    sp = mp.Space(p, J)
    nd = np.array([max(1, sp.valid_segment_count(s1[i], s2[i])) for i in range(n)], np.int32)
    # -- the C ABI through ctypes
    both = np.empty((2 * n, s1.shape[1]))
    both[0::2], both[1::2] = s1, s2
    abi_flags = eng.is_valid_batch(np.ascontiguousarray(both.T), A.LAYOUT_OMPL)["flags"]
    abi_ok, abi_first = eng.check_motion_batch(np.ascontiguousarray(s1.T), np.ascontiguousarray(s2.T), nd)
    eng.close()
    assert np.array_equal(got["valid_single"], abi_flags)
    assert np.array_equal(got["valid_batch"], abi_flags)
    assert np.array_equal(got["motion2"], abi_ok)          # the two-argument overload said "valid" for every edge
    assert np.array_equal(got["motion3"], abi_ok)
    assert np.array_equal(got["motions_ok"], abi_ok) and np.array_equal(got["motions_first"], abi_first)
    assert 0 < abi_ok.sum() < n, "the case must hold valid and invalid motions"
    assert list(got["counters"]) == [2 * int(abi_ok.sum()), 2 * int(n - abi_ok.sum()), 0, 0]
    # -- lastValid: fraction (first_invalid - 1) / nd, state = interpolate(s1, s2, fraction)
    for i in range(n):
        if abi_ok[i]:
            assert got["last_frac"][i] == -1.0 and np.array_equal(got["last_state"][i], s1[i])   # untouched
        else:
            frac = (abi_first[i] - 1) / nd[i]
            assert got["last_frac"][i] == frac
            assert np.allclose(got["last_state"][i], sp.interpolate(s1[i], s2[i], frac), rtol=0, atol=1e-15)
    # -- the oracle
    ref = O.eval_batch(m, boxes, p, np.ascontiguousarray(both.T), A.LAYOUT_OMPL)
    off_band = ~ref["band"].astype(bool)
    assert np.array_equal(got["valid_single"][off_band], ref["flags"][off_band])
    ok_o, first_o, band_o = O.check_motion_batch(m, boxes, p, np.ascontiguousarray(s1.T), np.ascontiguousarray(s2.T), nd)
    clear = ~band_o.astype(bool)
    assert np.array_equal(got["motion2"][clear], ok_o[clear])
    assert np.array_equal(got["motions_first"][clear], first_o[clear])
    if name == "c1":
        assert got["valid_single"][0] == 1 and got["valid_single"][1] == 1 and got["motion2"][0] == 0   # through the wall


@pytest.mark.parametrize("case", list(cases()), ids=lambda c: c[0])
def test_adapter_clearance_is_the_signed_min_distance(case):
    """ompl::base::StateValidityChecker::clearance / isValid(state, dist) through base-class pointers
    (what a MaximizeClearanceValidStateSampler calls; the reference installs a valid-state-sampler allocator,
    motion_planning.h:168-171) = mvk_result::min_dist = the oracle's signed separation."""
    name, m, p, boxes, s1, s2 = case
    n = len(s1)
    eng = mp.Engine(m, boxes, p)
    cl, d = np.zeros(n), np.zeros(n)
    v = np.zeros(n, np.uint8)
    err = np.zeros(1, np.uint32)
    rc = check_lib().mvk_adapter_check_clearance(eng.ctx, m.n_joint, p_(s1), n, p_(cl), p_(v), p_(d), p_(err))
    eng.close()
    assert rc == 0 and err[0] == 0
    ref = O.eval_batch(m, boxes, p, np.ascontiguousarray(s1.T), A.LAYOUT_OMPL)
    assert np.array_equal(cl, d)                                   # one evaluation serves both virtuals
    off_band = ~ref["band"].astype(bool)
    assert np.array_equal(v[off_band], ref["flags"][off_band])
    if name == "mode0":                                            # collision stage off: nothing to be clear of
        assert np.all(np.isposinf(cl)) and np.all(np.isposinf(ref["min_dist"]))
    else:
        assert np.allclose(cl, ref["min_dist"], rtol=1e-5, atol=1e-12)
        assert (cl <= 0).any() or name != "c1"
        assert (cl > 0).any()


def test_adapter_turns_errors_into_invalid_without_throwing():
    m = mp.make_model("hydrus", 4)
    p = mp.default_params(m, A.MODE_JOINTS_AND_BASE, A.SE2)
    eng = mp.Engine(m, mp.horizontal_gap_scene(), p)
    s1 = np.array([-1.0, 0.0, 0.0, 1.57, 1.57, 1.57])
    s2 = np.array([-1.2, 0.1, 0.1, 1.5, 1.5, 1.5])
    out = np.zeros(8, np.uint32)
    rc = check_lib().mvk_adapter_check_errors(eng.ctx, 3, p_(s1), p_(s2), p_(out))
    assert rc == 0, rc
    # isValid / checkMotion x2 answered "invalid", one error each, invalid motions counted, lastValid fraction 0, text kept
    assert list(out) == [0, 0, 0, 1, 2, 2, 1, 1]
    assert eng.is_valid_batch(s1.reshape(-1, 1), A.LAYOUT_OMPL)["flags"][0] == 1   # the ctx is usable again
    eng.close()
