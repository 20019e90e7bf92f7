#!/usr/bin/env python3
"""Fingerprint of the kernel sources: sha256 over csrc/mvk_kernels.hip, csrc/mvk_internal.h and include/mvk.h.

The PMC passes (run_profile.sh, pmc_full.sh) store it next to their counters at collection time; summarize.py copies
it into profiles/latest_pmc.json; bench.py compares it with the sources it runs from and refuses to derive
roofline.frac from counters of another kernel (a stale PMC made frac silently stale)."""
import hashlib
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ("motion_planning_amd/csrc/mvk_kernels.hip", "motion_planning_amd/csrc/mvk_internal.h", "include/mvk.h")


def kernel_fingerprint(root=ROOT):
    h = hashlib.sha256()
    for f in FILES:
        h.update(open(os.path.join(root, f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(kernel_fingerprint(sys.argv[1] if len(sys.argv) > 1 else ROOT))
