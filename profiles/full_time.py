#!/usr/bin/env python3
"""full path (distances, margins, statistics): time of full_kernel and far_kernel per 1M states, and how many states were far"""
import os, sys
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
    eng.set_profiling(True)
    d = torch.from_numpy(w.soa).cuda()
    md = torch.zeros(n, dtype=torch.float64, device="cuda")
    fl = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ts = []
    for i in range(4):
        eng.is_valid_batch_dev(d, w.layout, d_flags=fl, d_min_dist=md)
        torch.cuda.synchronize()
        if i >= 1:
            ts.append(eng.last_kernel_ms())
    t = np.array(ts)
    m = md.cpu().numpy()
    print(f"{name}: full_kernel {t[:, 0].mean():.2f} ms, far_kernel {t[:, 1].mean():.2f} ms; min_dist > 0.3: {(m > 0.3).mean():.3f}, > 0.5: {(m > 0.5).mean():.3f}, > 1.0: {(m > 1.0).mean():.3f}", flush=True)
    eng.close()
