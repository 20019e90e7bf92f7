#!/usr/bin/env python3
"""flag-path time of a HYDRUS quad batch (1M states, 40 boxes) -- occupancy experiments for the 4-link kernels"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import common
import motion_planning_amd as mp
n = 1 << 20
w = common.config2(n=n, n_links=4, n_boxes=40)
eng = mp.Engine(w.model, w.boxes, w.params)
eng.set_profiling(True)
d = torch.from_numpy(w.soa).cuda()
mask = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
ts = []
for i in range(12):
    eng.is_valid_mask_dev(d, w.layout, mask)
    torch.cuda.synchronize()
    if i >= 2:
        ts.append(eng.last_kernel_ms())
import numpy as np
t = np.array(ts)
print("hydrus-4 1M states: screen %.1f us, exact %.1f us" % (t[:, 0].mean() * 1e3, t[:, 1].mean() * 1e3))
