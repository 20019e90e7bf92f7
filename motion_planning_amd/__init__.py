"""motion_planning_amd -- MI355X-native batched state / motion validity engine.

The product is the C-ABI library built from motion_planning_amd/csrc
(include/mvk.h); this package is the ctypes plumbing used by tests and bench.py.
"""
from . import _abi as abi
from ._abi import (LAYOUT_OMPL, LAYOUT_POSE7, MODE_JOINTS_AND_BASE, MODE_ONLY_BASE, MODE_ONLY_JOINTS, SE2, SE3)
from . import path
from .engine import (Engine, Pipe, MvkError, Space, boxes_from_array, broadphase_query, far_grid_query, grid_bounds_query, sweep_states, default_params, horizontal_gap_scene, make_model,
                     vertical_gap_scene)

__all__ = ["abi", "Engine", "Pipe", "MvkError", "Space", "make_model", "default_params", "horizontal_gap_scene",
           "vertical_gap_scene", "boxes_from_array", "LAYOUT_OMPL", "LAYOUT_POSE7", "MODE_ONLY_JOINTS",
           "MODE_ONLY_BASE", "MODE_JOINTS_AND_BASE", "SE2", "SE3"]
