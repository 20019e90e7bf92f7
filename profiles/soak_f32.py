#!/usr/bin/env python3
"""The fp32 variant of the distance / margin outputs on the fuzz test's random scenes (plates, needles, a point; states
inside, around and far outside the scene), under many seeds: |d32 - d64| against the stated per-state tolerance
(mvk_f32_distance_tolerance), the margin enclosures against the fp64 margin, the flags outside the variant's band.
usage: python profiles/soak_f32.py [first] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import common
import motion_planning_amd as mp
from motion_planning_amd import _abi as A

DRAWS = [("hydrus", 4, 70), ("hydrus", 6, 300), ("dragon", 4, 1000), ("hydrus", 8, 600), ("dragon", 4, 120), ("hydrus", 6, 1000)]
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 16384
worst = 0.0
t0 = time.time()
for off in range(first, first + count):
    for kind, n_links, n_boxes in DRAWS:
        rng = np.random.default_rng(77 + 1000 * n_links + n_boxes + (7 if kind == "dragon" else 0) + 100003 * off)
        model = mp.make_model(kind, n_links)
        se3 = kind == "dragon" or n_boxes % 2 == 0
        params = mp.default_params(model, A.MODE_JOINTS_AND_BASE, A.SE3 if se3 else A.SE2)
        J = model.n_joint
        boxes = common.random_boxes(rng, n_boxes, [-3, -3, -2], [3, 3, 2], 0.02, 0.3, "so3" if se3 else "yaw")
        boxes[0:10, 7] = 0.0          # plates
        boxes[10:20, 8:10] = 1e-3     # needles
        boxes[20, 7:10] = 0.0         # a point
        st = np.zeros((7 + J, n))
        scale = np.where(np.arange(n) % 4 == 3, 3.0, 1.0)
        st[0:3] = rng.uniform(-2.5, 2.5, (3, n)) * scale
        st[3:7] = common.random_unit_quats(rng, n) if se3 else common.yaw_quats(rng.uniform(-np.pi, np.pi, n))
        st[7:] = rng.uniform(-1.5, 1.5, (J, n))
        eng = mp.Engine(model, boxes, params)
        ref = eng.eval_batch(st, A.LAYOUT_POSE7)
        d = torch.from_numpy(st).cuda()
        f = dict(flags=torch.zeros(n, dtype=torch.uint8, device="cuda"), band=torch.zeros(n, dtype=torch.uint8, device="cuda"),
                 min_dist=torch.zeros(n, dtype=torch.float32, device="cuda"), margin=torch.zeros(n, dtype=torch.float32, device="cuda"),
                 margin_err=torch.zeros(n, dtype=torch.float32, device="cuda"))
        eng.eval_batch_f32_dev(d, A.LAYOUT_POSE7, d_flags=f["flags"], d_min_dist=f["min_dist"], d_margin=f["margin"],
                               d_margin_err=f["margin_err"], d_band=f["band"])
        torch.cuda.synchronize()
        tol = np.array([eng.f32_distance_tolerance(x) for x in np.abs(st[0:3]).max(0)])
        eng.close()
        f = {k: v.cpu().numpy() for k, v in f.items()}
        fin = np.isfinite(ref["min_dist"])
        err = np.abs(f["min_dist"][fin].astype(np.float64) - ref["min_dist"][fin])
        ratio = float(np.max(err / tol[fin])) if fin.any() else 0.0
        worst = max(worst, ratio)
        e, m = f["margin_err"].astype(np.float64), f["margin"].astype(np.float64)
        ok = np.isfinite(e) & np.isfinite(ref["margin"])
        viol = np.abs(m[ok] - ref["margin"][ok]) > e[ok] + 4e-7 * np.maximum(np.abs(ref["margin"][ok]), 1.0)
        band = f["band"].astype(bool)
        flips = int(np.sum((f["flags"] != ref["flags"]) & ~band))
        bad = ratio > 1.0 or viol.any() or flips
        if bad:
            i = int(np.flatnonzero(fin)[np.argmax(err / tol[fin])])
            print(f"FAILED offset {off} {kind}-{n_links} {n_boxes} boxes: worst error / tolerance {ratio:.3f} at state {i} (d32 {f['min_dist'][i]!r}, "
                  f"d64 {ref['min_dist'][i]!r}, tol {tol[i]:.3g}); enclosure violations {int(viol.sum())}; flags off outside the band {flips}", flush=True)
            sys.exit(1)
    print(f"seed offset {off}: {len(DRAWS)} scenes x {n} states ok; worst |d32 - d64| / tolerance so far {worst:.3f} ({time.time() - t0:.0f} s)", flush=True)
print(f"{count} offsets x {len(DRAWS)} scenes: fp32 distances within their tolerance (worst ratio {worst:.3f}), every finite enclosure holds the fp64 margin, flags equal outside the band")
