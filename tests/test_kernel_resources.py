"""Register / scratch / LDS budgets of the built kernels, read from the compiler's own report
(-Rpass-analysis=kernel-resource-usage, written by csrc/Makefile next to the objects).  These are the numbers DESIGN.md 5
argues with: the flag path's dominant kernel must keep its occupancy without scratch, the model kernel must stay free
of scratch, the distance kernels must keep two waves per SIMD, and the latency kernels must not spill."""
import os
import re

import pytest

REPORT = os.path.join(os.path.dirname(__file__), "..", "motion_planning_amd", "lib", "obj", "kernel_resource_usage.txt")


def _kernels():
    if not os.path.exists(REPORT):
        pytest.skip("library not built here (python -c 'import __graft_entry__ as g; g.build()')")
    out, cur = {}, None
    for line in open(REPORT):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def _pick(kernels, *needles):
    hits = [v for k, v in kernels.items() if all(n in k for n in needles)]
    assert hits, needles
    return hits


def test_flag_path_kernels_keep_their_occupancy_without_scratch():
    k = _kernels()
    # screen_kernel<HYDRUS, 6, states, not the refine pass>: the headline kernel (c2k, c2)
    for r in _pick(k, "13screen_kernelILi0ELi6ELb0ELb0EE"):
        assert r["ScratchSize"] == 0 and r["Occupancy"] >= 3 and r["VGPRs Spill"] == 0, r
    for r in _pick(k, "12exact_kernelILi0ELi6E"):
        assert r["ScratchSize"] == 0 and r["Occupancy"] >= 3, r
    # every non-refine instantiation of the 4- and 6-link HYDRUS flag path (states and edges; the planner's fused
    # launches included) is free of scratch at three waves per SIMD (round 3: HYDRUS-4 went from four waves with 27-54
    # spilled VGPRs to three without: 104 -> 85 us per 1 M states, profiles/r03/occupancy_variants.txt)
    for nl in (4, 6):
        for name in (f"13screen_kernelILi0ELi{nl}ELb0ELb0EE", f"13screen_kernelILi0ELi{nl}ELb1ELb0EE", f"12fused_kernelILi0ELi{nl}E"):
            for r in _pick(k, name):
                assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0 and r["Occupancy"] >= 3, (name, r)
    # HYDRUS-8 and DRAGON need 172-180 VGPRs; at three waves per SIMD (168) they spill a little, and that is the faster
    # choice: two waves without spills measured 172 -> 204 us (c5) and 207 -> 242 us (c3) per 1 M states.  Bounded here
    # so that a regression shows.
    for name, cap in (("13screen_kernelILi0ELi8ELb0ELb0EE", 64), ("13screen_kernelILi0ELi8ELb1ELb0EE", 32),
                      ("13screen_kernelILi1ELi4ELb0ELb0EE", 8), ("13screen_kernelILi1ELi4ELb1ELb0EE", 8)):
        for r in _pick(k, name):
            assert r["VGPRs Spill"] <= cap and r["Occupancy"] >= 3, (name, r)


def test_model_kernel_has_no_scratch_and_distance_kernels_keep_two_waves():
    k = _kernels()
    for r in _pick(k, "12model_kernelI"):
        assert r["ScratchSize"] == 0 and r["VGPRs Spill"] == 0, r
    for name in ("15distance_kernelI", "12avoid_kernelI"):
        for r in _pick(k, name):
            assert r["Occupancy"] >= 2 and r["LDS Size"] <= 80 * 1024, r     # two blocks per CU within 160 KB of LDS


def test_latency_kernels_do_not_spill_vector_registers():
    k = _kernels()
    for name in ("17flag_small_kernelI", "21distance_small_kernelI", "17full_small_kernelI", "18avoid_small_kernelI", "18fused_split_kernelI"):
        for r in _pick(k, name):
            assert r["VGPRs Spill"] == 0, (name, r)
