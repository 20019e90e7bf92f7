#!/usr/bin/env python3
"""flag path: fp64 against the fp32 variant (MVK_PRECISION_F32_REFINED) per workload: ms per 1M-state call
(torch events over 10 calls), the library's per-stage split, the share of states refined in fp64, mask agreement"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import common
import motion_planning_amd as mp
from motion_planning_amd import _abi as A
from test_gpu_f32 import debug_bounds
n = 1 << 20
for name in sys.argv[1:] or ["c2", "c2k", "c3", "c5"]:
    w = {"c2": common.config2, "c2k": common.config2k, "c3": common.config3, "c5": common.config5}[name](n=n)
    eng = mp.Engine(w.model, w.boxes, w.params)
    eng.set_profiling(True)
    d = torch.from_numpy(w.soa).cuda()
    masks = {}
    line = []
    for prec, tag in ((A.PRECISION_F64, "fp64"), (A.PRECISION_F32_REFINED, "fp32")):
        eng.set_flag_precision(prec)
        m = torch.zeros(n // 64, dtype=torch.int64, device="cuda")
        for _ in range(3):
            eng.is_valid_mask_dev(d, w.layout, m)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(10):
            eng.is_valid_mask_dev(d, w.layout, m)
        b.record()
        torch.cuda.synchronize()
        k = eng.last_kernel_ms()
        masks[tag] = m.cpu().numpy()
        line.append(f"{tag} {a.elapsed_time(b) / 10:.4f} ms/call (screen {k[0]:.4f} + exact {k[1]:.4f})")
    lo, hi, cls = debug_bounds(eng, w.soa, w.layout)
    diff = int(np.unpackbits((masks["fp64"] ^ masks["fp32"]).view(np.uint8)).sum())
    print(f"{name}: " + "; ".join(line) + f"; refined {np.mean(cls == -1):.4f}, stable {np.mean(cls == 1):.3f}, unstable {np.mean(cls == 0):.3f}; mask differences {diff}", flush=True)
    eng.close()
