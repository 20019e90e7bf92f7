#!/usr/bin/env python3
"""Host round-trip latency of the small batched calls the planner makes (GPU box)."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import common
import motion_planning_amd as mp

model, params, boxes, start, goal = common.config1()
eng = mp.Engine(model, boxes, params)
rng = np.random.default_rng(0)
# the first ~0.1 s of GPU work in a process runs several times slower (clocks, lazy loading): warm up first
_w1 = np.tile(start[:, None], (1, 64)); _w2 = _w1 + 0.05
for _ in range(1500):
    eng.check_motion_batch(_w1, _w2, np.full(64, 41, np.int32))
for E in ([int(x) for x in sys.argv[1:]] or [1, 64, 256, 1024]):
    s1 = np.tile(start[:, None], (1, E)) + rng.normal(scale=0.05, size=(6, E))
    s2 = s1 + rng.normal(scale=0.1, size=(6, E))
    nd = np.full(E, 41, np.int32)
    for _ in range(20):
        eng.check_motion_batch(s1, s2, nd)
    t0 = time.perf_counter()
    reps = 300
    for _ in range(reps):
        eng.check_motion_batch(s1, s2, nd)
    dt = (time.perf_counter() - t0) / reps
    eng.set_profiling(True)
    eng.check_motion_batch(s1, s2, nd)
    k1, k2 = eng.last_kernel_ms()
    eng.set_profiling(False)
    print(f"edges {E:5d} x 41 steps: {dt*1e6:8.1f} us per call (incl. ctypes), screen {k1*1e3:.1f} us, exact {k2*1e3:.1f} us")
eng.close()
