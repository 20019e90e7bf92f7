#!/usr/bin/env python3
"""distance paths with and without the far grid (MVK_NO_FAR_GRID=1): full path per 1 M states, avoidance rows per
256 k, on the bench workloads and on the c2k scene seen from its rim (most primitives beyond the second grid)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import common
import motion_planning_amd as mp
n = 1 << 20
na = 1 << 18

def timed(fn, reps=3):
    fn(); torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps

for name in sys.argv[1:] or ["c2k", "c2k_rim", "c3", "c5", "c2"]:
    w = {"c2": common.config2, "c2k": common.config2k, "c2k_rim": common.config2k, "c3": common.config3, "c5": common.config5}[name](n=n)
    soa = w.soa
    if name == "c2k_rim":
        soa = soa.copy()
        soa[2] = np.where(np.arange(n) % 2 == 0, 3.2, -1.2)
    d = torch.from_numpy(soa).cuda()
    P = w.model.n_prim
    line = f"{name}:"
    for far in (True, False):
        if far:
            os.environ.pop("MVK_NO_FAR_GRID", None)
        else:
            os.environ["MVK_NO_FAR_GRID"] = "1"
        t0 = time.perf_counter()
        eng = mp.Engine(w.model, w.boxes, w.params)
        t_create = (time.perf_counter() - t0) * 1e3
        md = torch.zeros(n, dtype=torch.float64, device="cuda")
        fl = torch.zeros(n, dtype=torch.uint8, device="cuda")
        full = timed(lambda: eng.is_valid_batch_dev(d, w.layout, d_flags=fl, d_min_dist=md))
        da = d[:, :na].contiguous()
        outs = dict(dist=torch.zeros(P, na, dtype=torch.float64, device="cuda"), ok=torch.zeros(na, dtype=torch.uint8, device="cuda"))
        av = timed(lambda: eng.collision_avoidance_batch_dev(da, w.layout, outs))
        s1 = np.ascontiguousarray(soa[:, :1])
        for _ in range(20):
            eng.collision_avoidance_batch(s1, w.layout)
        t0 = time.perf_counter()
        for _ in range(100):
            eng.collision_avoidance_batch(s1, w.layout)
        one = (time.perf_counter() - t0) / 100 * 1e6
        line += f"  [{'far grid' if far else 'sweep   '}] full {full:.3f} ms/1M, rows {av:.3f} ms/256k, rows of one state {one:.0f} us, create {t_create:.0f} ms;"
        eng.close()
    print(line, flush=True)
