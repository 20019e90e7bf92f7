#!/usr/bin/env python3
"""avoidance rows on small device batches: avoid_small_kernel (a wave per state) against avoid_kernel (a lane per
state), time per call by batch size (device pointers, stream synchronised)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import common
import motion_planning_amd as mp
sizes = [int(x) for x in os.environ.get("SIZES", "1,8,64,512,2048,4096,8192,16384").split(",")]
for name in sys.argv[1:] or ["c2k", "c3"]:
    w = {"c2": common.config2, "c2k": common.config2k, "c3": common.config3, "c5": common.config5}[name](n=1 << 14)
    d = torch.from_numpy(w.soa).cuda()
    P, W = w.model.n_prim, 6 + w.model.n_joint
    for n in sizes:
        line = f"{name} n={n:6d}:"
        for small in (True, False):
            os.environ["MVK_NO_SMALL_FULL"] = "0" if small else "1"
            os.environ["MVK_SMALL_FULL_MAX"] = str(1 << 20)
            eng = mp.Engine(w.model, w.boxes, w.params)
            dn = d[:, :n].contiguous()
            outs = dict(dist=torch.zeros(P, n, dtype=torch.float64, device="cuda"), box=torch.zeros(P, n, dtype=torch.int32, device="cuda"),
                        p_env=torch.zeros(3 * P, n, dtype=torch.float64, device="cuda"), p_robot=torch.zeros(3 * P, n, dtype=torch.float64, device="cuda"),
                        row=torch.zeros(W * P, n, dtype=torch.float64, device="cuda"), lb=torch.zeros(P, n, dtype=torch.float64, device="cuda"),
                        ok=torch.zeros(n, dtype=torch.uint8, device="cuda"))
            ts = []
            for i in range(5):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                eng.collision_avoidance_batch_dev(dn, w.layout, outs)
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            line += f"  {'wave per state' if small else 'lane per state'} {1e6 * min(ts[1:]):8.1f} us"
            eng.close()
        print(line, flush=True)
