#!/usr/bin/env python3
"""flag-only host calls of a few states: flag_small_kernel (a block of two waves per state) against fused_kernel
(a lane per state), wall time per call by batch size"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import common
import motion_planning_amd as mp
for name in sys.argv[1:] or ["c2k", "c3"]:
    w = {"c2": common.config2, "c2k": common.config2k, "c3": common.config3, "c5": common.config5}[name](n=1024)
    for n in (1, 4, 16, 64, 128, 256, 512, 1024):
        line = f"{name} n={n:5d}:"
        for small in (True, False):
            os.environ["MVK_FLAG_SMALL_MAX"] = "1024" if small else "1"
            os.environ["MVK_NO_SMALL_FULL"] = "0" if small else "1"
            eng = mp.Engine(w.model, w.boxes, w.params)
            s = np.ascontiguousarray(w.soa[:, :n])
            for _ in range(20):
                eng.is_valid_batch(s, w.layout)
            t0 = time.perf_counter()
            for _ in range(200):
                eng.is_valid_batch(s, w.layout)
            line += f"  {'block per state' if small else 'lane per state '} {(time.perf_counter() - t0) / 200 * 1e6:6.1f} us"
            eng.close()
        print(line, flush=True)
