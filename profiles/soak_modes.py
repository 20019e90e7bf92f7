#!/usr/bin/env python3
"""tests/test_gate_overrides.py's GPU test -- every (robot, planning mode, motion type) the reference has, each with the
four (check_stability, check_collision) overrides, flag path, full path, edges and mvk_is_valid against the oracle -- on
other random scenes and states: seeds first .. first + count - 1, one process, stops at the first failure.
usage: python profiles/soak_modes.py [first] [count]"""
import os, sys, time, traceback
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gate_overrides as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1
count = int(sys.argv[2]) if len(sys.argv) > 2 else 10
orig = T.workload
t0 = time.time()
for off in range(first, first + count):
    T.workload = lambda kind, mode, mt, n=1500, seed=0, _o=off: orig(kind, mode, mt, n, seed=1000 * _o)
    for kind, mode, mt in T.COMBOS:
        try:
            T.test_gpu_gate_overrides_vs_oracle(kind, mode, mt)
        except Exception:
            traceback.print_exc()
            print(f"FAILED at seed {off}: {kind} mode {mode} motion type {mt}", flush=True)
            sys.exit(1)
    print(f"seed {off}: {len(T.COMBOS)} (robot, mode, type) combinations x 4 gate settings ok ({time.time() - t0:.0f} s)", flush=True)
print(f"{count} seeds: every mode / motion type / gate override agrees with the oracle")
