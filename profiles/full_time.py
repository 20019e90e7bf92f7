#!/usr/bin/env python3
"""full path (distances, margins, statistics): time of full_kernel per 1M states (MVK_NO_DISTANCE_GRID=1 in the
environment: without the distance grid) and the collision-avoidance rows per 256k states"""
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
    na = 1 << 18
    P, W = w.model.n_prim, 6 + w.model.n_joint
    da = d[:, :na].contiguous()
    outs = dict(dist=torch.zeros(P, na, dtype=torch.float64, device="cuda"), box=torch.zeros(P, na, dtype=torch.int32, device="cuda"),
                p_env=torch.zeros(3 * P, na, dtype=torch.float64, device="cuda"), p_robot=torch.zeros(3 * P, na, dtype=torch.float64, device="cuda"),
                row=torch.zeros(W * P, na, dtype=torch.float64, device="cuda"), lb=torch.zeros(P, na, dtype=torch.float64, device="cuda"),
                ok=torch.zeros(na, dtype=torch.uint8, device="cuda"))
    ta = []
    for i in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.collision_avoidance_batch_dev(da, w.layout, outs)
        torch.cuda.synchronize()
        ta.append(time.perf_counter() - t0)
    print(f"{name}: model_kernel {t[:, 0].mean():.3f} + distance_kernel {t[:, 1].mean():.3f} ms per 1M; avoidance rows {1e3 * min(ta[1:]):.2f} ms per 256k; min_dist > 0.3: {(m > 0.3).mean():.3f}, > 0.5: {(m > 0.5).mean():.3f}, > 1.0: {(m > 1.0).mean():.3f}", flush=True)
    eng.close()
