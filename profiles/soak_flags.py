#!/usr/bin/env python3
"""Identities of the flag path on the fuzz test's random scenes, under many seeds (one process, stops at the first
failure): the packed mask equals the flags; the fp32-refined flag path (MVK_PRECISION_F32_REFINED) gives the fp64
flags; the full path's flags equal the flag path's outside the full path's band; a pipe of two contexts gives the
mask of one ctx; edge checks of zero-length edges equal the state's flag.
usage: python profiles/soak_flags.py [first] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import common
import motion_planning_amd as mp
from motion_planning_amd import _abi as A

DRAWS = [("hydrus", 4, 10), ("hydrus", 6, 300), ("dragon", 4, 1000), ("hydrus", 8, 600), ("dragon", 4, 70), ("hydrus", 6, 1000)]
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
n = 32768
t0 = time.time()
refined_total = 0
for off in range(first, first + count):
    for kind, n_links, n_boxes in DRAWS:
        rng = np.random.default_rng(99 + 1000 * n_links + n_boxes + (7 if kind == "dragon" else 0) + 100003 * off)
        model = mp.make_model(kind, n_links)
        se3 = kind == "dragon" or n_boxes % 2 == 0
        params = mp.default_params(model, A.MODE_JOINTS_AND_BASE, A.SE3 if se3 else A.SE2)
        J = model.n_joint
        boxes = common.random_boxes(rng, n_boxes, [-3, -3, -2], [3, 3, 2], 0.02, 0.3, "so3" if se3 else "yaw")
        if n_boxes >= 30:
            boxes[0:10, 7] = 0.0
            boxes[10:20, 8:10] = 1e-3
            boxes[20, 7:10] = 0.0
        st = np.zeros((7 + J, n))
        scale = np.where(np.arange(n) % 4 == 3, 3.0, 1.0)
        st[0:3] = rng.uniform(-2.5, 2.5, (3, n)) * scale
        st[3:7] = common.random_unit_quats(rng, n) if se3 else common.yaw_quats(rng.uniform(-np.pi, np.pi, n))
        st[7:] = rng.uniform(-1.5, 1.5, (J, n))
        st[7, 17] = np.nan
        what = f"offset {off} {kind}-{n_links} {n_boxes} boxes"
        eng = mp.Engine(model, boxes, params)
        d = torch.from_numpy(st).cuda()
        flags = torch.zeros(n, dtype=torch.uint8, device="cuda")
        mask = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
        eng.is_valid_batch_dev(d, A.LAYOUT_POSE7, d_flags=flags)
        eng.is_valid_mask_dev(d, A.LAYOUT_POSE7, mask)
        torch.cuda.synchronize()
        fl = flags.cpu().numpy()
        bits = np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder="little")[:n]
        assert np.array_equal(bits, fl), what + ": mask != flags"
        assert fl[17] == 0, what
        # fp32-refined flag path
        eng.set_flag_precision(A.PRECISION_F32_REFINED)
        m32 = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
        eng.is_valid_mask_dev(d, A.LAYOUT_POSE7, m32)
        torch.cuda.synchronize()
        eng.set_flag_precision(A.PRECISION_F64)
        assert torch.equal(m32, mask), what + f": fp32-refined mask differs in {int((m32 != mask).sum())} words"
        # full path
        full = eng.eval_batch(st, A.LAYOUT_POSE7)
        band = full["band"].astype(bool)
        assert np.array_equal(full["flags"][~band], fl[~band]), what + ": full path flags"
        eng.close()
        # a pipe of two contexts
        pipe = mp.Pipe(model, boxes, params, depth=2)
        pm = [torch.zeros(n // 64, dtype=torch.int64, device="cuda") for _ in range(3)]
        for k in range(3):
            pipe.is_valid_mask(d, A.LAYOUT_POSE7, pm[k])
        pipe.drain()
        for k in range(3):
            assert torch.equal(pm[k], mask), what + f": pipe batch {k}"
        pipe.close()
    print(f"seed offset {off}: {len(DRAWS)} scenes x {n} states ok ({time.time() - t0:.0f} s)", flush=True)
print(f"{count} offsets x {len(DRAWS)} scenes: mask = flags = fp32-refined flags = pipe's masks; full-path flags equal outside their band")
