#!/usr/bin/env python3
"""RRT-Connect gap passes on the GPU checker under many seeds, every plan verified edge by edge with the CPU oracle
(config c1: HYDRUS quad through the horizontal gap; the 6-link variant of tests/test_gpu_planner.py is left to the suite),
and the sweep entry point on random index ranges against the materialised states.
usage: python profiles/soak_planner.py [first seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import common
import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A
from motion_planning_amd import planner as PL

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 40
model, params, boxes, start, goal = common.config1()
eng = mp.Engine(model, boxes, params)
sp = O.Space(params, 3)
t0 = time.time()
solved, ms = 0, []
for seed in range(first, first + count):
    wp, info = PL.plan_rrt_connect(eng, start, goal, time_limit_s=30.0, seed=seed, batch=256)
    if wp is None:
        print(f"seed {seed}: no plan within the limit ({info})", flush=True)
        continue
    solved += 1
    ms.append(info.get("solve_ms", float("nan")))
    nd = np.array([max(1, sp.valid_segment_count(wp[i], wp[i + 1])) for i in range(len(wp) - 1)], np.int32)
    ok, _, band = O.check_motion_batch(model, boxes, params, wp[:-1].T.copy(), wp[1:].T.copy(), nd)
    flags = O.eval_batch(model, boxes, params, wp.T.copy(), A.LAYOUT_OMPL)
    if not np.all(ok | band) or not np.all(flags["flags"].astype(bool) | flags["band"].astype(bool)):
        print(f"FAILED seed {seed}: the oracle rejects an edge or a waypoint of the GPU checker's plan", flush=True)
        sys.exit(1)
    dense = PL.densify(params, 3, wp)
    ref = O.eval_batch(model, boxes, params, dense.T.copy(), A.LAYOUT_OMPL, threads=8)
    if not np.all(ref["flags"].astype(bool) | ref["band"].astype(bool)):
        print(f"FAILED seed {seed}: a state of the densified path is invalid per the oracle", flush=True)
        sys.exit(1)
eng.close()
print(f"planner: {solved} of {count} seeds solved, every plan valid per the oracle; median solve {np.median(ms):.2f} ms, max {np.max(ms):.2f} ({time.time() - t0:.0f} s)", flush=True)

# the sweep on random index ranges
rng = np.random.default_rng(first)
w = common.config5(n=8)
eng = mp.Engine(w.model, w.boxes, w.params)
for k in range(count):
    first_index = int(rng.integers(0, 1 << 40))
    n = int(rng.integers(1, 200000))
    seed = int(rng.integers(0, 1 << 62))
    pose = np.r_[rng.uniform(-1, 1, 3), common.random_unit_quats(rng, 1)[:, 0]]
    mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    eng.sweep_mask_dev(pose, first_index, n, seed, mask)
    torch.cuda.synchronize()
    st = eng.sweep_states(pose, first_index, n, seed)
    ref = eng.is_valid_batch(st, A.LAYOUT_POSE7)["flags"]
    bits = np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder="little")
    if not np.array_equal(bits[:n], ref) or bits[n:].any():
        print(f"FAILED sweep first {first_index} n {n} seed {seed}", flush=True)
        sys.exit(1)
eng.close()
print(f"sweep: {count} random index ranges equal the materialised states, tail bits clear")
