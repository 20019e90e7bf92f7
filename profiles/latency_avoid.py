#!/usr/bin/env python3
"""latency of the collision-avoidance rows for the one state a QP iteration of the differential-kinematics planner
asks about (collision_avoidance.cpp:102-277), and for small batches"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import common
import motion_planning_amd as mp
for name, w in (("c2k (1000 boxes)", common.config2k(n=4096)), ("c3 dragon (1000 boxes)", common.config3(n=4096)), ("c2 (100 boxes)", common.config2(n=4096)), ("c2k again", common.config2k(n=4096))):
    eng = mp.Engine(w.model, w.boxes, w.params)
    for n in (1, 8, 64, 512, 4096):
        s = np.ascontiguousarray(w.soa[:, :n])
        for _ in range(5):
            eng.collision_avoidance_batch(s, w.layout)
        t0 = time.perf_counter()
        for _ in range(30):
            eng.collision_avoidance_batch(s, w.layout)
        print(f"{name}: avoidance rows, {n:5d} states per call: {(time.perf_counter() - t0) / 30 * 1e6:.0f} us per call", flush=True)
    eng.close()
