#!/usr/bin/env python3
"""More draws of tests/test_gpu_fuzz.py's random scenes (every kernel family against the oracle) than the test suite
runs: seed offsets first .. first + count - 1, in one process, stopping at the first failure.
usage: python profiles/soak_fuzz.py [first] [count]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as F

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 8
t0 = time.time()
for off in range(first, first + count):
    os.environ["MVK_FUZZ_SEED_OFFSET"] = str(off)
    for kind, n_links, n_boxes in F.DRAWS:
        try:
            F.test_random_scene_all_kernel_families(kind, n_links, n_boxes)
        except Exception:
            traceback.print_exc()
            print(f"FAILED at seed offset {off}: {kind} {n_links} links, {n_boxes} boxes", flush=True)
            sys.exit(1)
    print(f"seed offset {off}: {len(F.DRAWS)} scenes ok ({time.time() - t0:.0f} s)", flush=True)
print(f"{count} offsets x {len(F.DRAWS)} scenes: every kernel family agrees with the oracle")
