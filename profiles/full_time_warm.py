#!/usr/bin/env python3
"""full path (model_kernel + distance_kernel: flags, exact minimum distance, margins, statistics) per 1 M states in steady
state: 0.3 s of launches first (GPU clocks up, as bench.py's pre-warm), then 50 launches between two events.
usage: python profiles/full_time_warm.py [workloads...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import common
import motion_planning_amd as mp
n = 1 << 20
for name in sys.argv[1:] or ["c2", "c2k", "c3", "c5"]:
    w = {"c2": common.config2, "c2k": common.config2k, "c3": common.config3, "c5": common.config5}[name](n=n)
    eng = mp.Engine(w.model, w.boxes, w.params)
    d = torch.from_numpy(w.soa).cuda()
    md = torch.zeros(n, dtype=torch.float64, device="cuda")
    mg = torch.zeros(n, dtype=torch.float64, device="cuda")
    fl = torch.zeros(n, dtype=torch.uint8, device="cuda")
    run = lambda: eng.is_valid_batch_dev(d, w.layout, d_flags=fl, d_min_dist=md, d_margin=mg)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(4):
            run()
        torch.cuda.synchronize()
    res = []
    for rnd in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(50):
            run()
        b.record()
        torch.cuda.synchronize()
        res.append(a.elapsed_time(b) / 50)
    eng.set_profiling(True)
    ks = []
    for _ in range(5):
        run(); torch.cuda.synchronize(); ks.append(eng.last_kernel_ms())
    k = np.array(ks)
    print(f"{name}: full path {min(res):.3f} ms per 1 M states (rounds {', '.join(f'{x:.3f}' for x in res)}); model {k[:, 0].mean():.3f} + distance {k[:, 1].mean():.3f} ms", flush=True)
    eng.close()
