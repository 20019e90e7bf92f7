#!/usr/bin/env python3
"""full path on small device batches: distance_small_kernel (a wave per state) against distance_kernel (a lane per
state), kernel time by batch size -- where the library's switch (kSmallFullMax) belongs"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import common
import motion_planning_amd as mp
for name in sys.argv[1:] or ["c2k", "c3"]:
    w = {"c2": common.config2, "c2k": common.config2k, "c3": common.config3, "c5": common.config5}[name](n=1 << 16)
    d = torch.from_numpy(w.soa).cuda()
    for n in (1, 64, 512, 2048, 4096, 8192, 16384, 32768, 65536):
        line = f"{name} n={n:6d}:"
        for small in (True, False):
            os.environ["MVK_NO_SMALL_FULL"] = "0" if small else "1"
            os.environ["MVK_SMALL_FULL_MAX"] = str(1 << 20)
            eng = mp.Engine(w.model, w.boxes, w.params)
            eng.set_profiling(True)
            md = torch.zeros(n, dtype=torch.float64, device="cuda")
            fl = torch.zeros(n, dtype=torch.uint8, device="cuda")
            dn = d[:, :n].contiguous()
            ts = []
            for i in range(5):
                eng.is_valid_batch_dev(dn, w.layout, d_flags=fl, d_min_dist=md)
                torch.cuda.synchronize()
                if i >= 2:
                    ts.append(eng.last_kernel_ms()[1])
            line += f"  {'wave per state' if small else 'lane per state'} {1e3 * np.mean(ts):8.1f} us"
            eng.close()
        print(line, flush=True)
