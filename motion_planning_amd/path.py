"""ctypes plumbing for the path post-processing entry points of include/mvk.h."""
import ctypes as C

import numpy as np

from . import _abi as A
from .engine import MvkError, _check


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def root_to_cog(model, state):
    lib = A.load()
    s = np.ascontiguousarray(state, np.float64)
    out = np.empty_like(s)
    _check(lib, lib.mvk_root_to_cog(C.byref(model), _p(s), _p(out)))
    return out


def cog_to_root(model, state):
    lib = A.load()
    s = np.ascontiguousarray(state, np.float64)
    out = np.empty_like(s)
    _check(lib, lib.mvk_cog_to_root(C.byref(model), _p(s), _p(out)))
    return out


def smooth(root_states, filter_rate=0.1):
    """root_states: [n, 7+J].  ContinuousPathGenerator::discretePathSmoothing (parity unpinned, include/mvk.h)."""
    lib = A.load()
    s = np.ascontiguousarray(root_states, np.float64)
    n, D = s.shape
    m = lib.mvk_path_smooth(_p(s), n, D - 7, float(filter_rate), None, 0)
    if m < 0:
        _check(lib, m)
    out = np.empty((m, D))
    _check(lib, min(0, lib.mvk_path_smooth(_p(s), n, D - 7, float(filter_rate), _p(out), m)))
    return out


def resample(cog_states, angular_rate=1.0, seg_diff_thresh=0.2):
    """cog_states: [n, 7+J].  ContinuousPathGenerator::discretePathResampling."""
    lib = A.load()
    s = np.ascontiguousarray(cog_states, np.float64)
    n, D = s.shape
    cap = 64 * n + 1024
    out = np.empty((cap, D))
    m = lib.mvk_path_resample(_p(s), n, D - 7, float(angular_rate), float(seg_diff_thresh), _p(out), cap)
    if m < 0:
        _check(lib, m)
    return out[:m].copy()


class Spline:
    """ContinuousPathGenerator::calcContinuousPath + BsplineRos."""

    def __init__(self, cog_states, period, degree=5):
        self.lib = A.load()
        s = np.ascontiguousarray(cog_states, np.float64)
        n, D = s.shape
        self.h = self.lib.mvk_spline_create(_p(s), n, D - 7, float(period), int(degree))
        if not self.h:
            raise MvkError(self.lib.mvk_last_error().decode())
        self.dim = self.lib.mvk_spline_dim(self.h)

    def close(self):
        if getattr(self, "h", None):
            self.lib.mvk_spline_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def eval(self, t, derive=0):
        t = np.ascontiguousarray(np.atleast_1d(t), np.float64)
        out = np.empty((len(t), self.dim))
        _check(self.lib, self.lib.mvk_spline_eval(self.h, _p(t), len(t), int(derive), _p(out)))
        return out

    def eval_dev(self, d_t, derive=0, device=0, stream=None):
        """d_t: torch float64 tensor on the device; returns [dim, nt] torch tensor."""
        import torch
        nt = d_t.numel()
        out = torch.empty((self.dim, nt), dtype=torch.float64, device=d_t.device)
        st = torch.cuda.current_stream() if stream is None else stream
        _check(self.lib, self.lib.mvk_spline_eval_dev(self.h, device, C.c_void_p(d_t.data_ptr()), nt, int(derive),
                                                      C.c_void_p(out.data_ptr()), nt, C.c_void_p(st.cuda_stream)))
        return out
