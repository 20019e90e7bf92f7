"""The C-ABI library loads and exports every symbol include/mvk.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import motion_planning_amd as mp
from motion_planning_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols(header="mvk.h"):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mvk_[a-z0-9_]+)\s*\(", text)))


def test_every_declared_symbol_is_exported_and_bound():
    lib = A.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"libmvk.so does not export {n}"
        assert n in A.PROTOTYPES, f"{n} has no ctypes prototype"
    assert sorted(A.PROTOTYPES) == names


def test_every_exported_symbol_is_declared_in_a_header():
    """no extern "C" export of the release library outside include/ (mvk.h = the boundary,
    mvk_debug.h = diagnostics)"""
    import subprocess
    lib = A.load()
    dbg = declared_symbols("mvk_debug.h")
    assert sorted(A.DEBUG_PROTOTYPES) == dbg
    for n in dbg:
        assert hasattr(lib, n), f"libmvk.so does not export {n}"
    out = subprocess.run(["nm", "-D", "--defined-only", A.lib_path()], capture_output=True, text=True, check=True).stdout
    exported = sorted({ln.split()[-1] for ln in out.splitlines() if re.search(r" T mvk_[a-z0-9_]+$", ln)})
    assert exported == sorted(declared_symbols() + dbg), set(exported) ^ set(declared_symbols() + dbg)


def test_struct_layouts_match():
    lib = A.load()
    assert lib.mvk_model_sizeof() == C.sizeof(A.Model)
    assert lib.mvk_params_sizeof() == C.sizeof(A.Params)
    assert lib.mvk_abi_version() == 1


def test_model_factories_and_argument_errors():
    lib = A.load()
    m = A.Model()
    assert lib.mvk_model_hydrus(2, C.byref(m)) == A.ERR_INVALID_ARG
    assert b"n_links" in lib.mvk_last_error()
    for n in (4, 6, 8):
        m = mp.make_model("hydrus", n)
        assert (m.n_seg, m.n_joint, m.n_rotor, m.n_prim, m.wrench_dof) == (n, n - 1, n, 2 * n, 4)
        assert abs(m.link_length - 0.6) < 1e-15          # hydrus.yaml:31
        assert abs(m.prim_dim[1][0] - 0.2) < 1e-15       # protector radius, hydrus.yaml:32
    d = mp.make_model("dragon", 4)
    assert (d.n_seg, d.n_joint, d.n_rotor, d.n_prim, d.wrench_dof, d.has_gimbal) == (7, 6, 4, 15, 3, 1)
    assert abs(d.edf_max_tilt - 0.1) < 1e-15             # dragon_quad.yaml:56
    p = mp.default_params(d, A.MODE_JOINTS_AND_BASE, A.SE3)
    assert (p.validity_resolution, p.valid_segment_count_factor, p.joint_subspace_weight) == (0.03, 20, 0.001)
    assert list(p.pos_low) == [-2.0, -2.2, -2.2] and list(p.pos_high) == [5.0, 2.2, 2.2]


@pytest.mark.skipif(A.load().mvk_device_count() > 0, reason="box has a GPU")
def test_no_device_means_loud_failure_not_fallback():
    lib = A.load()
    m = mp.make_model("hydrus", 4)
    p = mp.default_params(m, 2, 0)
    with pytest.raises(mp.MvkError) as e:
        mp.Engine(m, np.zeros((0, 10)), p)
    assert "no HIP device" in str(e.value) or "not gfx950" in str(e.value)
    with pytest.raises(mp.MvkError) as e:          # the pipe owns contexts: the same refusal, nothing half-built
        mp.Pipe(m, np.zeros((0, 10)), p, depth=2)
    assert "no HIP device" in str(e.value) or "not gfx950" in str(e.value)
    out = C.c_void_p()
    empty = np.zeros((0, 10))   # (the pipe returns the code of the mvk_create that failed, not a generic one)
    rc = lib.mvk_pipe_create(C.byref(m), empty.ctypes.data_as(A.PB), 0, C.byref(p), 0, 2, C.byref(out))
    assert rc == A.ERR_NO_DEVICE and not out.value
    with pytest.raises(mp.MvkError):
        mp.Pipe(m, np.zeros((0, 10)), p, depth=0)
    assert lib.mvk_pipe_depth(None) == 0 and not lib.mvk_pipe_ctx(None, 0) and not lib.mvk_pipe_stream(None, 0)
    lib.mvk_pipe_destroy(None)


def test_gap_worlds_literal_numbers():
    """squeeze_navigation/src/planner/sampling_based_method.cpp:84-112 and :116-197"""
    h = mp.horizontal_gap_scene()
    want = np.array([[0.15, 1.55, 0, 0, 0, 0, 1, 0.3, 2.5, 10], [0.15, -1.55, 0, 0, 0, 0, 1, 0.3, 2.5, 10],
                     [1.0, 2.5, 0, 0, 0, 0, 1, 8, 0.6, 10], [1.0, -2.5, 0, 0, 0, 0, 1, 8, 0.6, 10]])
    assert np.array_equal(h, want)
    # quirk Q2: right wall keeps the LEFT wall's x size but is positioned with gap_right_width
    h2 = mp.horizontal_gap_scene(0.1, 0.4, 0.3, 0.5)
    assert h2[1, 0] == 0.1 + 0.25 and h2[1, 7] == 0.3
    v = mp.vertical_gap_scene(0.5, 0.5, 0.7, True, 0.6, 2.0)   # gap.yaml:2-4
    assert v.shape == (10, 10)
    assert np.allclose(v[0], [5.25, 0, 0.7, 0, 0, 0, 1, 10, 10, 0.05])
    assert np.allclose(v[3], [0, -5.25, 0.7, 0, 0, 0, 1, 10, 10, 0.05])
    assert np.allclose(v[4], [0, 0, 1.3, 0, 0, 0, 1, 10, 10, 0.05])     # ceiling
    assert np.allclose(v[5], [0, 0, 0.0, 0, 0, 0, 1, 10, 10, 0.05])     # ground
    assert np.allclose(v[6], [1.0, 0, 0.35, 0, 0, 0, 1, 0.05, 2.0, 0.7])
    assert np.allclose(v[9], [0, -1.0, 0.35, 0, 0, 0, 1, 2.0, 0.05, 0.7])
    assert mp.vertical_gap_scene(0.5, 0.5, 0.7, False).shape == (4, 10)


def test_headers_compile_as_c_and_cpp(tmp_path):
    """include/mvk.h is valid C11; the OMPL-shaped adapter compiles against the OMPL look-alike stub."""
    import subprocess
    inc = os.path.join(ROOT, "include")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(inc, "mvk.h")])
    src = tmp_path / "adapter_use.cpp"
    src.write_text('''
#define MVK_OMPL_STUB
#include "mvk_ompl_adapter.hpp"
int use(mvk_ctx* ctx) {
  auto si = std::make_shared<ompl::base::SpaceInformation>();
  mvk::StateValidityChecker svc(si, ctx, 3);
  mvk::MotionValidator mv(si, ctx, 3);
  ompl::base::SE2StateSpace::StateType b1, b2; double j1[3] = {0, 0, 0}, j2[3] = {1, 1, 1};
  ompl::base::RealVectorStateSpace::StateType r1, r2; r1.values = j1; r2.values = j2;
  ompl::base::State* c1[2] = {&b1, &r1}; ompl::base::State* c2[2] = {&b2, &r2};
  ompl::base::CompoundState s1, s2; s1.components = c1; s2.components = c2;
  std::pair<ompl::base::State*, double> last(&s1, 0.0);
  const ompl::base::State* many[2] = {&s1, &s2}; std::uint8_t out[2];
  svc.isValid(many, 2, out);
  double dist = 0.0;
  return svc.isValid(&s1) + svc.isValid(&s1, dist) + mv.checkMotion(&s1, &s2) + mv.checkMotion(&s1, &s2, last);
}
''')
    subprocess.check_call(["g++", "-std=c++17", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, str(src)])


def test_integration_examples_compile_and_link(tmp_path):
    """the calls INTEGRATION.md shows (pipe, multi-GPU exchange, fp32 outputs), as C against mvk.h and libmvk.so"""
    import subprocess
    libdir = os.path.join(ROOT, "motion_planning_amd", "lib")
    if not os.path.exists(os.path.join(libdir, "libmvk.so")):
        pytest.skip("library not built here")
    out = tmp_path / "libexamples.so"
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-shared", "-fPIC", "-Wl,--no-undefined", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "examples", "integration_snippets.c"), "-L", libdir, "-lmvk", "-o", str(out)])


def test_adapter_check_library_is_built_and_exports_its_entry_points():
    """the runnable form of the adapter (csrc/mvk_adapter_check.cpp); it is RUN by tests/test_gpu_adapter.py"""
    A.load()
    lib = C.CDLL(os.path.join(os.path.dirname(A.lib_path()), "libmvk_adapter_check.so"))
    assert hasattr(lib, "mvk_adapter_check_run") and hasattr(lib, "mvk_adapter_check_errors")
