#!/usr/bin/env python3
"""the fp32 variant of the distance / margin outputs (mvk_eval_batch_f32_dev) next to the fp64 full path: kernel times
per 1M states (library events) and wall time per call"""
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
    mg = torch.zeros(n, dtype=torch.float64, device="cuda")
    fl = torch.zeros(n, dtype=torch.uint8, device="cuda")
    md32 = torch.zeros(n, dtype=torch.float32, device="cuda")
    mg32 = torch.zeros(n, dtype=torch.float32, device="cuda")
    me32 = torch.zeros(n, dtype=torch.float32, device="cuda")
    bd = torch.zeros(n, dtype=torch.uint8, device="cuda")
    res = {}
    for kind, fn in (("f64", lambda: eng.is_valid_batch_dev(d, w.layout, d_flags=fl, d_min_dist=md, d_margin=mg)),
                     ("f32", lambda: eng.eval_batch_f32_dev(d, w.layout, d_flags=fl, d_min_dist=md32, d_margin=mg32, d_margin_err=me32, d_band=bd))):
        ts, wall = [], []
        for i in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            wall.append(time.perf_counter() - t0)
            if i >= 1:
                ts.append(eng.last_kernel_ms())
        res[kind] = (np.array(ts).mean(0), 1e3 * min(wall[1:]))
    err = (md32.double() - md).abs()
    err = err[torch.isfinite(md)]
    print(f"{name}: f64 model {res['f64'][0][0]:.3f} + distance {res['f64'][0][1]:.3f} ms (call {res['f64'][1]:.3f}); "
          f"f32 model {res['f32'][0][0]:.3f} + distance {res['f32'][0][1]:.3f} ms (call {res['f32'][1]:.3f}); max |d32 - d64| {err.max().item():.3g} m", flush=True)
    eng.close()
