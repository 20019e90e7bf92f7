#!/usr/bin/env python3
"""a 20 000-box scene: time of mvk_create (three grids built on the host), full-path time with and without the distance
grid, and that both give the same distances bit for bit"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch  # noqa
import common
import motion_planning_amd as mp
rng = np.random.default_rng(5)
w = common.config2k(n=1 << 17)
w.boxes = common.random_boxes(rng, 20000, [-10, -10, -4], [10, 10, 4], 0.02, 0.3, "so3")
w.soa[0] = rng.uniform(-9, 9, w.soa.shape[1]); w.soa[1] = rng.uniform(-9, 9, w.soa.shape[1]); w.soa[2] = rng.uniform(-3, 3, w.soa.shape[1])
out = []
for off in ("0", "1"):
    os.environ["MVK_NO_DISTANCE_GRID"] = off
    t0 = time.perf_counter()
    eng = mp.Engine(w.model, w.boxes, w.params)
    t1 = time.perf_counter()
    eng.set_profiling(True)
    d = torch.from_numpy(w.soa).cuda()
    n = d.shape[1]
    md = torch.zeros(n, dtype=torch.float64, device="cuda"); fl = torch.zeros(n, dtype=torch.uint8, device="cuda")
    for _ in range(2):
        eng.is_valid_batch_dev(d, w.layout, d_flags=fl, d_min_dist=md)
        torch.cuda.synchronize()
    mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device="cuda")
    for _ in range(3):
        eng.is_valid_mask_dev(d, w.layout, mask)
        torch.cuda.synchronize()
    k = eng.last_kernel_ms()
    print(f"flag path: screen {1e3 * k[0]:.0f} us + exact {1e3 * k[1]:.0f} us per {n} states = {n / (k[0] + k[1]) * 1e3:.3g} checks/s")
    eng.is_valid_batch_dev(d, w.layout, d_flags=fl, d_min_dist=md)
    torch.cuda.synchronize()
    print(f"distance grid {'off' if off == '1' else 'on'}: mvk_create {1e3 * (t1 - t0):.0f} ms, full path (model + distance kernel) {sum(eng.last_kernel_ms()):.2f} ms per {n} states, valid {int(fl.sum())}", flush=True)
    out.append(md.cpu().numpy())
    eng.close()
print("identical distances:", np.array_equal(out[0], out[1]))
