#!/usr/bin/env python3
"""RRT-Connect gap pass (c1), 10 seeds at batch 256 with MVK_PLAN_TRACE=1: where an iteration's wall time goes"""
import os, sys
os.environ["MVK_PLAN_TRACE"] = "1"
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import common
import motion_planning_amd as mp
from motion_planning_amd import planner as PL
model, params, boxes, start, goal = common.config1()
eng = mp.Engine(model, boxes, params)
batch = int(sys.argv[1]) if len(sys.argv) > 1 else 256
for _ in range(3):
    PL.plan_rrt_connect(eng, start, goal, time_limit_s=20.0, seed=99, batch=batch)   # warm-up
print("---", flush=True)
ms = []
for seed in range(1, 11):
    wp, info = PL.plan_rrt_connect(eng, start, goal, time_limit_s=20.0, seed=seed, batch=batch)
    ms.append(info["solve_ms"])
print(f"batch {batch}: median {np.median(ms):.2f} ms, mean {np.mean(ms):.2f}, per iteration (seed 2) see trace", flush=True)
eng.close()
