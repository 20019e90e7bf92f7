import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import torch
import common
import motion_planning_amd as mp
model, params, boxes, start, goal = common.config1()
eng = mp.Engine(model, boxes, params)
rng = np.random.default_rng(0)
for E in (256, 512):
    s1 = np.tile(start[:, None], (1, E)) + rng.normal(scale=0.05, size=(6, E))
    s2 = s1 + rng.normal(scale=0.1, size=(6, E))
    for nd in (np.full(E, 41, np.int32), 41):
        for _ in range(20):
            eng.check_motion_batch(s1, s2, nd)
        t0 = time.perf_counter()
        for _ in range(300):
            eng.check_motion_batch(s1, s2, nd)
        dt = (time.perf_counter() - t0) / 300
        eng.set_profiling(True)
        ks = []
        for _ in range(10):
            eng.check_motion_batch(s1, s2, nd); ks.append(eng.last_kernel_ms())
        eng.set_profiling(False)
        k = np.array(ks).mean(0)
        print(f"E {E} nd {'array' if not np.isscalar(nd) else 'uniform'}: {dt*1e6:.1f} us per call, screen {k[0]*1e3:.1f}, exact {k[1]*1e3:.1f}")
eng.close()
