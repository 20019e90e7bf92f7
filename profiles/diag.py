#!/usr/bin/env python3
"""Kernel-time breakdown on the GPU box: stability-only / collision-only / both, flag vs full kernel, edges.
usage: python profiles/diag.py [workloads...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402

import common  # noqa: E402
import motion_planning_amd as mp  # noqa: E402


def timeit(fn, reps=10):
    fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


PROF_NAMES = {0: "lanes", 1: "live", 2: "sure_hit", 3: "undecided", 4: "free(no maybe)", 5: "screen wave-iters", 6: "screen lane-iters",
              8: "exact L1 wave-iters", 9: "exact L2 wave-iters", 10: "exact L2 lane-iters", 11: "narrow wave-calls",
              12: "narrow lane-calls", 13: "exact hits", 15: "phase-2 waves"}


def prof_counters(w, d, mask):
    """event counters of one launch (diagnostic library built with -DMVK_PROF only)"""
    import ctypes as C
    lib = mp.abi.load()
    if not hasattr(lib, "mvk_debug_prof"):
        return
    buf = (C.c_ulonglong * 48)()
    eng = mp.Engine(w.model, w.boxes, w.params)
    for _ in range(3):  # warm: buffers allocated and touched, code resident
        eng.is_valid_mask_dev(d, w.layout, mask)
    torch.cuda.synchronize()
    lib.mvk_debug_prof(None, 1)
    eng.is_valid_mask_dev(d, w.layout, mask)
    torch.cuda.synchronize()
    lib.mvk_debug_prof(buf, 1)
    eng.close()
    print("   counters: " + ", ".join(f"{PROF_NAMES[k]}={buf[k]}" for k in sorted(PROF_NAMES)), flush=True)
    waves = max(int(buf[15]), 1)
    stages = ["entry->queue", "fetch+frame", "level 1", "level 2"]
    if os.environ.get("MVK_DIAG_SCREEN"):   # library built with -DMVK_PROFS: the same slots hold the screen kernel's stages
        stages = ["fetch+walk 1", "walk 2 per-segment part", "walk 2 pops", "tail"]
    print("   exact_kernel stage clocks (us, mean over waves / max): " +
          ", ".join(f"{nm} {buf[16 + 2 * i] / waves / 100:.2f}/{buf[17 + 2 * i] / 100:.2f}" for i, nm in enumerate(stages)) +
          f"; first entry -> last exit {(buf[41] - buf[40]) / 100:.2f}, waves {waves}", flush=True)


def main():
    names = sys.argv[1:] or ["c2", "c2k", "c3", "c5"]
    dev = torch.device("cuda", 0)
    n = 1 << 20
    for name in names:
        w = {"c2": common.config2, "c2k": common.config2k, "c3": common.config3, "c5": common.config5,
             "c2se3": lambda n: common.config2(n=n, orient="so3")}[name](n=n)
        d = torch.from_numpy(w.soa).to(dev)
        mask = torch.zeros(n // 64, dtype=torch.int64, device=dev)
        md = torch.zeros(n, dtype=torch.float64, device=dev)
        flags = torch.zeros(n, dtype=torch.uint8, device=dev)
        row = [name]
        for stab, coll in ((1, 0), (0, 1), (1, 1)):
            p = type(w.params).from_buffer_copy(w.params)
            p.check_stability, p.check_collision = stab, coll
            eng = mp.Engine(w.model, w.boxes, p)
            ms = timeit(lambda: eng.is_valid_mask_dev(d, w.layout, mask))
            row.append(f"stab={stab} coll={coll}: {ms:.3f} ms")
            eng.close()
        eng = mp.Engine(w.model, w.boxes, w.params)
        nf = 1 << 16
        ms = timeit(lambda: eng.is_valid_batch_dev(d[:, :nf], w.layout, d_flags=flags[:nf], d_min_dist=md[:nf]), reps=3)
        row.append(f"full kernel: {ms / nf * n:.2f} ms per 1M")
        eng.close()
        print(" | ".join(row), flush=True)
        prof_counters(w, d, mask)
    # c4 edges
    model, params, boxes, s1, s2, nd = common.config4(n_edges=1 << 16)
    eng = mp.Engine(model, boxes, params)
    d1, d2 = torch.from_numpy(s1).to(dev), torch.from_numpy(s2).to(dev)
    ok = torch.zeros(s1.shape[1], dtype=torch.uint8, device=dev)
    first = torch.zeros(s1.shape[1], dtype=torch.int32, device=dev)
    ms = timeit(lambda: eng.check_motion_batch_dev(d1, d2, nd, ok, first))
    E = s1.shape[1]
    print(f"c4 edges: {ms:.3f} ms for {E} edges x {nd} steps = {E / ms * 1e3:.4g} edges/s = {E * nd / ms * 1e3:.4g} checks/s; ok fraction {ok.float().mean().item():.3f}")
    eng.close()


if __name__ == "__main__":
    main()
