#!/usr/bin/env python3
"""event counters of the distance kernels (library built with -DMVK_PROF, MVK_LIB=<it>): per (state, primitive),
how many lanes the distance grid could not answer, how many candidate lists overflowed, mean list length"""
import ctypes as C
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import common
import motion_planning_amd as mp
lib = mp.abi.load()
n = 1 << 18
for name in sys.argv[1:] or ["c2", "c2k", "c3", "c5"]:
    w = {"c2": common.config2, "c2k": common.config2k, "c3": common.config3, "c5": common.config5}[name](n=n)
    eng = mp.Engine(w.model, w.boxes, w.params)
    d = torch.from_numpy(w.soa).cuda()
    md = torch.zeros(n, dtype=torch.float64, device="cuda")
    fl = torch.zeros(n, dtype=torch.uint8, device="cuda")
    buf = (C.c_ulonglong * 48)()
    lib.mvk_debug_prof(None, 1)
    eng.is_valid_batch_dev(d, w.layout, d_flags=fl, d_min_dist=md)
    torch.cuda.synchronize()
    lib.mvk_debug_prof(buf, 1)
    b = [int(x) for x in buf]
    print(f"{name} full : lane-prims {b[30]}, swept {b[31]} ({b[31] / max(b[30], 1):.4f}), overflow {b[32]} ({b[32] / max(b[30], 1):.4f}), mean candidates {b[33] / max(b[30], 1):.2f}")
    wv = max(b[28], 1)
    print(f"{name} full : per wave, us: model update {b[24] / wv / 100:.1f}, collision {b[25] / wv / 100:.1f} = first walk {b[20] / wv / 100:.1f} + candidates {b[21] / wv / 100:.1f}"
          f" + drain {b[22] / wv / 100:.1f} (of which exact stage {b[23] / wv / 100:.1f}); exact pairs per state {b[34] / n:.2f}, exact rounds per wave {b[35] / wv:.1f}; waves {wv}")
    print(f"{name} full : flat box drain: rounds of 64 pairs per wave {b[26] / wv:.1f}, separating-axis bounds per state {b[27] / n:.2f}")
    P = w.model.n_prim
    outs = dict(dist=torch.zeros(P, n, dtype=torch.float64, device="cuda"), ok=torch.zeros(n, dtype=torch.uint8, device="cuda"))
    eng.collision_avoidance_batch_dev(d, w.layout, outs)
    torch.cuda.synchronize()
    lib.mvk_debug_prof(buf, 1)
    b = [int(x) for x in buf]
    wv = max(b[47], 1)
    print(f"{name} avoid: per wave, us: candidates {b[40] / wv / 100:.1f}, drain {b[41] / wv / 100:.1f} (gaps {b[43] / wv / 100:.1f}, first exact {b[44] / wv / 100:.1f}), rows + stores {b[42] / wv / 100:.1f}; waves {wv}")
    print(f"{name} avoid: after round 0 and the centre bound: {b[45] / max(b[36], 1):.2f} candidates per lane-prim left; exact closest-point pairs beyond round 0: {b[46] / max(b[36], 1):.3f} per lane-prim")
    print(f"{name} avoid: lane-prims {b[36]}, swept {b[37]} ({b[37] / max(b[36], 1):.4f}), overflow {b[38]} ({b[38] / max(b[36], 1):.4f}), mean candidates {b[39] / max(b[36], 1):.2f}", flush=True)
    eng.close()
