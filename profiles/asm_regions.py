#!/usr/bin/env python3
"""Static instruction counts of one kernel in a device-only assembly listing (hipcc -S --cuda-device-only):
the whole kernel and every backward-branch region (loops).  usage: asm_regions.py <file.s> <mangled-name-substring>"""
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and sys.argv[2] in l.split(":")[0])
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
body = lines[start + 1:end]
labels = {}
for i, l in enumerate(body):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        labels[m.group(1)] = i


def count(a, b):
    c = dict(valu=0, f64=0, f32=0, salu=0, lds=0, vmem=0)
    for l in body[a:b]:
        op = l.strip().split(" ")[0].split("\t")[0]
        if op.startswith("v_"):
            c["valu"] += 1
            c["f64"] += "f64" in op
            c["f32"] += "f32" in op
        elif op.startswith("s_"):
            c["salu"] += 1
        elif op.startswith("ds_"):
            c["lds"] += 1
        elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
            c["vmem"] += 1
    return c


print("kernel", len(body), "lines", count(0, len(body)))
for i, l in enumerate(body):
    m = re.search(r"s_(?:c)?branch\w*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        a = labels[m.group(1)]
        print(f"loop {m.group(1)} lines {a}-{i}", count(a, i))
