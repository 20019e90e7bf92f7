#!/usr/bin/env python3
"""one ctx on one stream against mvk_pipe_* (depth 2, 3) per workload: wall clock per step over 400 steps, drained on both
sides, three rounds in alternation (clocks up first)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import common
import motion_planning_amd as mp
n = 1 << 20
for name in sys.argv[1:] or ["c2k", "c2", "c3"]:
    w = {"c2": common.config2, "c2k": common.config2k, "c3": common.config3}[name](n=n)
    d = torch.from_numpy(w.soa).cuda()
    words = n // 64
    eng = mp.Engine(w.model, w.boxes, w.params)
    pipes = {dep: mp.Pipe(w.model, w.boxes, w.params, depth=dep) for dep in (2, 3)}
    m1 = torch.zeros(words, dtype=torch.int64, device="cuda")
    pm = [torch.zeros(words, dtype=torch.int64, device="cuda") for _ in range(6)]
    s = torch.cuda.Stream()

    def one(k):
        eng.is_valid_mask_dev(d, w.layout, m1, stream=s)

    def timed(fn, drain, steps=400):
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < 0.25:
            for k in range(16):
                fn(k)
            drain()
        t0 = time.perf_counter()
        for k in range(steps):
            fn(k)
        drain()
        return (time.perf_counter() - t0) / steps * 1e3
    res = {"one": [], 2: [], 3: []}
    for rnd in range(3):
        res["one"].append(timed(one, torch.cuda.synchronize))
        for dep in (2, 3):
            p = pipes[dep]
            res[dep].append(timed(lambda k: p.is_valid_mask(d, w.layout, pm[k % (2 * dep)]), p.drain))
    print(f"{name}: one ctx {min(res['one']):.4f} ms per step; pipe depth 2 {min(res[2]):.4f}; depth 3 {min(res[3]):.4f}   (all rounds: {res})", flush=True)
    eng.close()
    for p in pipes.values():
        p.close()
