#!/usr/bin/env python3
"""How much of a flag-path launch is lost to waves whose lanes are mostly decided by the stability gate before the
collision stage: the same batch with the stable states first (dense waves in the collision stage) against the batch as
sampled.  An upper bound on what compacting the live lanes inside the kernel could gain.
usage: python profiles/live_sort_probe.py [workloads...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import common
import motion_planning_amd as mp


def timeit(fn, reps=200):
    for _ in range(300):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


dev = torch.device("cuda", 0)
n = 1 << 20
for name in sys.argv[1:] or ["c3", "c2k", "c5"]:
    w = {"c2": common.config2, "c2k": common.config2k, "c3": common.config3, "c5": common.config5}[name](n=n)
    eng = mp.Engine(w.model, w.boxes, w.params)
    d = torch.from_numpy(w.soa).to(dev)
    margin = torch.zeros(n, dtype=torch.float64, device=dev)
    flags = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng.is_valid_batch_dev(d, w.layout, d_flags=flags, d_margin=margin)
    torch.cuda.synchronize()
    stable = (margin >= 0).cpu().numpy()
    order = np.argsort(~stable, kind="stable")
    ds = torch.from_numpy(np.ascontiguousarray(w.soa[:, order])).to(dev)
    mask = torch.zeros(n // 64, dtype=torch.int64, device=dev)
    t0 = timeit(lambda: eng.is_valid_mask_dev(d, w.layout, mask))
    t1 = timeit(lambda: eng.is_valid_mask_dev(ds, w.layout, mask))
    print(f"{name}: stable fraction {stable.mean():.3f}; as sampled {t0 * 1e3:.1f} us per launch, stable states first {t1 * 1e3:.1f} us", flush=True)
    eng.close()
