#!/usr/bin/env python3
"""latency of single-state calls: mvk_is_valid (full kernel, distance + margin) vs a batch of one through the flag path.
After an idle spell (the start of the process, or host-only work between two kinds of call) the GPU serves the first
~50 ms of small launches several times slower -- its clocks ramp up -- which is why c2k is measured twice here and why
bench.py's single_calls warms every kind of call for 0.2 s before timing it."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa
import common
import motion_planning_amd as mp
from motion_planning_amd import _abi as A
for name, w in (("c2k (1000 boxes)", common.config2k(n=64)), ("c2 (100 boxes)", common.config2(n=64)), ("c2k again", common.config2k(n=64))):
    eng = mp.Engine(w.model, w.boxes, w.params)
    s = np.ascontiguousarray(w.soa[:, :1])
    for fn, label in ((lambda: eng.is_valid_batch(s, w.layout), "flag path, batch of one"),
                      (lambda: eng.is_valid(s[:, 0]), "mvk_is_valid (full kernel: verdict, distance, margin)"),
                      (lambda: eng.eval_batch(s, w.layout), "full kernel with every detail output, batch of one")):
        for _ in range(10):
            fn()
        t0 = time.perf_counter()
        for _ in range(100):
            fn()
        print(f"{name}: {label}: {(time.perf_counter() - t0) / 100 * 1e6:.0f} us per call")
    eng.close()
