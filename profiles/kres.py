#!/usr/bin/env python3
"""Per-kernel resource usage (VGPRs, AGPRs, scratch, occupancy, LDS) from the compiler's
-Rpass-analysis=kernel-resource-usage remarks that csrc/Makefile keeps in lib/obj/kernel_resource_usage.txt."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def table(path=None):
    path = path or os.path.join(ROOT, "motion_planning_amd", "lib", "obj", "kernel_resource_usage.txt")
    cur, name = {}, None
    for line in open(path):
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1)
            cur[name] = {}
            continue
        m = re.search(r"remark:\s+([A-Za-z][^:]*): (\d+)", line)
        if m and name:
            cur[name][m.group(1).strip()] = int(m.group(2))
    out = {}
    for n, v in cur.items():
        dn = subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
        dn = dn.replace("void mvk::(anonymous namespace)::", "")
        dn = re.sub(r"\(mvk::EvalArgs.*|\(int\*.*|\(mvk::StatsPartial.*|\(int const\*.*", "", dn)
        out[dn] = v
    return out


if __name__ == "__main__":
    for dn, v in table(sys.argv[1] if len(sys.argv) > 1 else None).items():
        keep = {k: v[k] for k in v if k.split()[0] in ("VGPRs", "AGPRs", "ScratchSize", "Occupancy", "LDS", "SGPRs")}
        print(f"{dn:44s}", keep)
