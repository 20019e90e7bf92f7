#!/usr/bin/env python3
"""mvk_set_scene wall time per workload scene (the four grids are built on the host), with the distance grids' trace"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["MVK_DISTANCE_GRID_TRACE"] = "1"
import common
import motion_planning_amd as mp
for name, f in [("c2", common.config2), ("c2k", common.config2k), ("c3", common.config3), ("c5", common.config5)]:
    w = f(n=64)
    eng = mp.Engine(w.model, w.boxes, w.params)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        eng.set_scene(w.boxes)
        ts.append(time.perf_counter() - t0)
    print(f"{name}: mvk_set_scene {1e3 * min(ts):.1f} ms ({len(w.boxes)} boxes)", flush=True)
    eng.close()
