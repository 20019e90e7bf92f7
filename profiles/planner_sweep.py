#!/usr/bin/env python3
"""RRT-Connect gap pass (config c1) on the GPU checker: solve time vs. batch size (GPU box)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import common
import motion_planning_amd as mp
from motion_planning_amd import planner as PL

model, params, boxes, start, goal = common.config1()
eng = mp.Engine(model, boxes, params)
for batch in (16, 64, 256, 1024):
    rows = []
    for seed in range(1, 11):
        wp, info = PL.plan_rrt_connect(eng, start, goal, time_limit_s=20.0, seed=seed, batch=batch)
        rows.append((info["solved"], info["solve_ms"], info["iterations"], info["edge_checks"]))
    ms = [r[1] for r in rows if r[0]]
    print(f"batch {batch:5d}: solved {len(ms)}/10, median {np.median(ms):8.1f} ms, mean {np.mean(ms):8.1f} ms, max {np.max(ms):8.1f} ms, "
          f"median iterations {np.median([r[2] for r in rows]):.0f}, median edge checks {np.median([r[3] for r in rows]):.0f}", flush=True)
eng.close()
