#!/usr/bin/env python3
"""flag-path time per 1M states of named workloads (h4: HYDRUS quad / 40 boxes, c2, c2k, c3, c5 sweep, c4 edges): steady state
(0.3 s of launches first), ms per step over 200 launches and the per-kernel split.  usage: flag_time.py [workload ...]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import common
import motion_planning_amd as mp
n = 1 << 20
for name in sys.argv[1:] or ["h4", "c2k", "c3", "c5"]:
    if name == "c4":
        model, params, boxes, s1, s2, nd = common.config4(n_edges=1 << 14)
        eng = mp.Engine(model, boxes, params)
        d1, d2 = torch.from_numpy(s1).cuda(), torch.from_numpy(s2).cuda()
        ok = torch.zeros(s1.shape[1], dtype=torch.uint8, device="cuda"); first = torch.zeros(s1.shape[1], dtype=torch.int32, device="cuda")
        run = lambda: eng.check_motion_batch_dev(d1, d2, nd, ok, first)
    else:
        w = {"h4": lambda n: common.config2(n=n, n_links=4, n_boxes=40), "c2": common.config2, "c2k": common.config2k,
             "c3": common.config3, "c5": lambda n: common.config5(n=64)}[name](n)
        eng = mp.Engine(w.model, w.boxes, w.params)
        mask = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
        if name == "c5":
            pose = np.array([0, 0, 0, 0, 0, 0, 1.0])
            run = lambda: eng.sweep_mask_dev(pose, 0, n, common.SEED0 + 5, mask)
        else:
            d = torch.from_numpy(w.soa).cuda()
            run = lambda: eng.is_valid_mask_dev(d, w.layout, mask)
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.3:
        for _ in range(16):
            run()
        torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(200):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / 200
    eng.set_profiling(True)
    ts = []
    for _ in range(8):
        run(); torch.cuda.synchronize(); ts.append(eng.last_kernel_ms())
    t = np.array(ts)
    print(f"{name}: {ms * 1e3:.1f} us per step; screen {t[:, 0].mean() * 1e3:.1f} + exact {t[:, 1].mean() * 1e3:.1f} us", flush=True)
    eng.close()
