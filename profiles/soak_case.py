#!/usr/bin/env python3
"""One draw of the fuzz test's scenes, the states whose margin differs from the oracle's printed with their details.
usage: python profiles/soak_case.py <offset> <kind> <n_links> <n_boxes>"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common
import motion_planning_amd as mp
import oracle_lib as O
from motion_planning_amd import _abi as A

off, kind, n_links, n_boxes = int(sys.argv[1]), sys.argv[2], int(sys.argv[3]), int(sys.argv[4])
rng = np.random.default_rng(1000 * n_links + n_boxes + (7 if kind == "dragon" else 0) + 100003 * off)
model = mp.make_model(kind, n_links)
se3 = kind == "dragon" or n_boxes % 2 == 0
params = mp.default_params(model, A.MODE_JOINTS_AND_BASE, A.SE3 if se3 else A.SE2)
J = model.n_joint
boxes = common.random_boxes(rng, n_boxes, [-3, -3, -2], [3, 3, 2], 0.02, 0.3, "so3" if se3 else "yaw")
if n_boxes >= 30:
    boxes[0:10, 7] = 0.0
    boxes[10:20, 8:10] = 1e-3
    boxes[20, 7:10] = 0.0
n = 2600
st = np.zeros((7 + J, n))
scale = np.where(np.arange(n) % 4 == 3, 3.0, 1.0)
st[0:3] = rng.uniform(-2.5, 2.5, (3, n)) * scale
st[3:7] = common.random_unit_quats(rng, n) if se3 else common.yaw_quats(rng.uniform(-np.pi, np.pi, n))
st[7:] = rng.uniform(-1.5, 1.5, (J, n))
st[8 if J > 1 else 7, 17] = np.nan
layout = A.LAYOUT_POSE7
ref = O.eval_batch(model, boxes, params, st, layout, threads=8)
eng = mp.Engine(model, boxes, params)
got = eng.eval_batch(st, layout)
fm = np.isfinite(ref["margin"])
bad = np.flatnonzero(fm & ~np.isclose(got["margin"], ref["margin"], rtol=1e-5, atol=1e-9))
print("states off:", bad)
np.set_printoptions(precision=17, linewidth=200)
for i in bad[:5]:
    print(f"state {i}: margin gpu {got['margin'][i]!r} oracle {ref['margin'][i]!r}  flags {got['flags'][i]} / {ref['flags'][i]}  band {got['band'][i]} / {ref['band'][i]}")
    for key in sorted(ref.keys()):
        if key in got and key not in ("margin", "flags", "band") and np.ndim(ref[key]) >= 1:
            a = np.asarray(got[key])[..., i] if np.asarray(got[key]).shape[-1] == n else None
            b = np.asarray(ref[key])[..., i] if np.asarray(ref[key]).shape[-1] == n else None
            if a is not None and b is not None:
                print("   ", key, a, b)
    print("    state", st[:, i])
    np.save(os.path.join(ROOT, "gpurun_out", f"soak_state_{off}_{i}.npy"), st[:, i])
eng.close()
