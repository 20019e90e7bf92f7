#!/usr/bin/env python3
"""bench.py -- state validity checks/sec on MI355X (BASELINE.json metric).

One "step" = one pass of the batched validity predicate over one resident batch of
the north star's target workload "c2k": HYDRUS 6-link, joints+base, 1,048,576 random
samples, 1000 box obstacles (FK + statics + collision), inputs already in HBM, flags
written as a packed bit mask.  The other single-GPU configurations of BASELINE.json
(c2: 100 boxes, c3: DRAGON + 1000 boxes, c5: HYDRUS 8-link C-space sweep) are timed
as legs of the same line ("workloads"), c4 as "edge_checks".
With N > 1 GPUs (one process per GPU, torch.distributed over RCCL) every rank
checks its own 1M-sample shard (weak scaling) and each step ends with the
all-gather of the packed masks + all-reduce of the valid count, the only
collectives the path needs.

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

# algorithmic work per check
F_MODEL = {("hydrus", 4): 2.4e3, ("hydrus", 6): 5.6e3, ("hydrus", 8): 11.3e3, ("dragon", 4): 4.7e3}
F_PAIR = 30.0
PEAK_FP64_VALU_TFLOPS = 78.6   # MI355X datasheet fp64 vector peak = 1024 SIMDs x 16 lanes x 2 flop x 2.4 GHz
PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
# VALU issue peak: a wave64 VALU instruction occupies its SIMD's 16 lanes for 4 cycles
# (256 CUs x 4 SIMDs x 2.4 GHz / 4 = 6.144e11 wave instructions / s); an fp64 FMA is one such instruction
PEAK_VALU_WAVE_INSTS_PER_S = 256 * 4 * 2.4e9 / 4


def prewarm(fn, seconds=0.25):
    """Out of an idle spell (host-only work: building a workload, the CPU baseline) the GPU runs its first ~50 ms of
    work 10-15 % slower (small launches: several times) while its clocks ramp up.  Every single-rank leg therefore
    keeps the GPU busy with its own call for a quarter of a second before it measures: steady-state rates, which is
    what a planner that checks batch after batch sees.  (The headline's timed region does the same with a fixed
    number of untimed steps, --prewarm, so that all ranks agree on the count.)"""
    import torch
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds:
        for _ in range(8):
            fn()
        torch.cuda.synchronize()


def cpu_baseline(w, budget_s=12.0):
    """The CPU checker (oracle, kind 'port') on a bounded sample of the same workload, all host cores
    the box gives this job (OMP_NUM_THREADS is set to the share on the GPU box)."""
    import oracle_lib as O
    cores = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(os.cpu_count() or 1, 16)
    cores = max(1, min(cores, 64))
    n = 20000
    soa = np.ascontiguousarray(w.soa[:, :n])
    t0 = time.perf_counter()
    O.flags_only(w.model, w.boxes, w.params, soa, w.layout, threads=cores)
    dt = time.perf_counter() - t0
    rate = n / dt
    # scale the sample so the timed run is ~budget_s of wall time, capped by the batch
    n2 = int(min(w.soa.shape[1], max(n, rate * budget_s)))
    soa = np.ascontiguousarray(w.soa[:, :n2])
    t0 = time.perf_counter()
    O.flags_only(w.model, w.boxes, w.params, soa, w.layout, threads=cores)
    dt = time.perf_counter() - t0
    # one thread as well (SURVEY.md 8(d) asks for both), on ~3 s of work
    n1 = int(min(n2, max(2000, rate / cores * 3.0)))
    soa1 = np.ascontiguousarray(w.soa[:, :n1])
    t0 = time.perf_counter()
    O.flags_only(w.model, w.boxes, w.params, soa1, w.layout, threads=1)
    dt1 = time.perf_counter() - t0
    return {"value": n2 / dt, "unit": "checks/s", "cores": cores, "kind": "port",
            "sample": f"first {n2} samples of the {w.name} batch, flag-only predicate, {cores} threads",
            "single_thread": {"value": n1 / dt1, "unit": "checks/s", "sample": f"first {n1} samples, 1 thread"}}


def rrt_connect_gap_pass(seeds=10, batch=256):
    """Second half of BASELINE.json's metric: RRT-Connect gap-pass solve time (config c1: HYDRUS quad through
    the default horizontal gap world) on the GPU checker; median over seeds of the search wall time.  The same
    planner is run once on the CPU checker (config 0 of BASELINE.json, the oracle behind the planner's callback
    form) for comparison."""
    import common
    import motion_planning_amd as mp
    import oracle_lib as O
    from motion_planning_amd import _abi as A
    from motion_planning_amd import planner as PL
    model, params, boxes, start, goal = common.config1()
    eng = mp.Engine(model, boxes, params)
    PL.plan_rrt_connect(eng, start, goal, time_limit_s=5.0, seed=99, batch=batch)   # warm-up (buffers, worker threads)
    runs = []
    for seed in range(1, seeds + 1):
        wp, info = PL.plan_rrt_connect(eng, start, goal, time_limit_s=20.0, seed=seed, batch=batch)
        runs.append(dict(seed=seed, solved=int(info["solved"]), solve_ms=info["solve_ms"], iterations=int(info["iterations"]),
                         edge_checks=int(info["edge_checks"]), state_checks_in_edges=int(info["edge_steps"]),
                         waypoints=int(info["n_waypoints"])))
    eng.close()
    ok = [r["solve_ms"] for r in runs if r["solved"]]
    out = {"config": f"c1: HYDRUS quad, joints+base SE2, horizontal gap world (4 boxes), {batch} extensions per iteration",
           "unit": "ms", "median_solve_ms": float(np.median(ok)) if ok else None, "mean_solve_ms": float(np.mean(ok)) if ok else None,
           "max_solve_ms": float(np.max(ok)) if ok else None, "solved": len(ok), "seeds": seeds, "runs": runs}

    def cs(soa):
        return O.eval_batch(model, boxes, params, soa, A.LAYOUT_OMPL)["flags"]

    def cm(s1, s2, nd):
        okk, first, _ = O.check_motion_batch(model, boxes, params, s1, s2, nd)
        return okk, first
    t0 = time.perf_counter()
    wp, info = PL.plan_rrt_connect_cb(params, model.n_joint, cs, cm, start, goal, time_limit_s=30.0, seed=1, batch=batch)
    out["cpu_checker"] = {"kind": "port", "seed": 1, "solved": int(info["solved"]), "solve_ms": info["solve_ms"],
                          "wall_ms": (time.perf_counter() - t0) * 1e3, "edge_checks": int(info["edge_checks"]),
                          "note": "same planner and seed on the single-thread CPU checker (oracle), 30 s limit"}
    return out


def edge_checks(dev):
    """config c4: batched MotionValidator edge checks, 65,536 edges x 64 interpolation steps, HYDRUS 6-link, 100 boxes."""
    import torch

    import common
    import motion_planning_amd as mp
    model, params, boxes, s1, s2, nd = common.config4(n_edges=1 << 16)
    eng = mp.Engine(model, boxes, params, device=dev.index)
    d1, d2 = torch.from_numpy(s1).to(dev), torch.from_numpy(s2).to(dev)
    E = s1.shape[1]
    ok = torch.zeros(E, dtype=torch.uint8, device=dev)
    first = torch.zeros(E, dtype=torch.int32, device=dev)
    prewarm(lambda: eng.check_motion_batch_dev(d1, d2, nd, ok, first))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 10
    a.record()
    for _ in range(reps):
        eng.check_motion_batch_dev(d1, d2, nd, ok, first)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / reps
    eng.close()
    return {"config": f"c4: HYDRUS 6-link, {E} edges x {nd} steps, 100 boxes, one GPU", "ms": ms,
            "edges_per_s": E / ms * 1e3, "checks_per_s": E * nd / ms * 1e3, "valid_edge_fraction": float(ok.float().mean().item())}


def distance_roofline(workload, kernel, n, ms):
    """VALU-issue roofline of a distance kernel, as roofline_of: executed wave instructions per launch (committed PMC
    pass, scaled to n states) / the time measured here / 6.144e11 per s.  None when no PMC pass is committed."""
    ent = pmc_traffic(workload + ".distance") or {}
    if ent.get("stale"):
        return {"bound": "valu_fp64", "kernel": kernel, "frac": None, "stale_pmc": ent["stale"]}
    k = ent.get(kernel)
    if not k:
        return None
    per = k.get("states_per_launch") or ent.get("states_per_launch")
    if not per:
        return {"bound": "valu_fp64", "kernel": kernel, "frac": None, "stale_pmc": "the PMC entry does not say how many states its launches held"}
    insts = k["valu_insts_per_launch"] * n / per
    return {"bound": "valu_fp64", "kernel": kernel, "frac": insts / (ms * 1e-3) / 6.144e11,
            "frac_definition": "executed VALU wave instructions per launch (PMC SQ_INSTS_VALU) / ms / 6.144e11 per s; ms here includes "
                               "model_kernel (3-5 % of the call)" if kernel == "distance_kernel" else
                               "executed VALU wave instructions per launch (PMC SQ_INSTS_VALU) / ms / 6.144e11 per s",
            "valu_wave_insts_per_launch": insts, "valu_busy_pmc": k["valu_busy"], "wait_any_frac_pmc": k["wait_any_frac"],
            "pmc_source": ent["source"]}


def distance_path(eng, w, d_soa, dev, workload=None):
    """the same batch through the distance kernels: full path (flags + exact minimum distance + margins + batch
    statistics) and the collision-avoidance rows (8(f)-3) on the first 262 144 states"""
    import torch
    n = d_soa.shape[1]
    fl = torch.zeros(n, dtype=torch.uint8, device=dev)
    md = torch.zeros(n, dtype=torch.float64, device=dev)
    mg = torch.zeros(n, dtype=torch.float64, device=dev)

    def timed(fn, reps=4):
        prewarm(fn)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps):
            fn()
        b.record()
        torch.cuda.synchronize()
        return a.elapsed_time(b) / reps
    full_ms = timed(lambda: eng.is_valid_batch_dev(d_soa, w.layout, d_flags=fl, d_min_dist=md, d_margin=mg))
    stats = eng.reduce_stats()
    na = min(n, 1 << 18)
    P, W = w.model.n_prim, 6 + w.model.n_joint
    da = d_soa[:, :na].contiguous()
    outs = dict(dist=torch.zeros(P, na, dtype=torch.float64, device=dev), box=torch.zeros(P, na, dtype=torch.int32, device=dev),
                p_env=torch.zeros(3 * P, na, dtype=torch.float64, device=dev), p_robot=torch.zeros(3 * P, na, dtype=torch.float64, device=dev),
                row=torch.zeros(W * P, na, dtype=torch.float64, device=dev), lb=torch.zeros(P, na, dtype=torch.float64, device=dev),
                ok=torch.zeros(na, dtype=torch.uint8, device=dev))
    avoid_ms = timed(lambda: eng.collision_avoidance_batch_dev(da, w.layout, outs))
    # the fp32 variant of the same outputs (mvk_eval_batch_f32_dev): its rate, its stated tolerance, what was observed
    f = dict(fl=torch.zeros(n, dtype=torch.uint8, device=dev), bd=torch.zeros(n, dtype=torch.uint8, device=dev),
             md=torch.zeros(n, dtype=torch.float32, device=dev), mg=torch.zeros(n, dtype=torch.float32, device=dev),
             me=torch.zeros(n, dtype=torch.float32, device=dev))
    f32_ms = timed(lambda: eng.eval_batch_f32_dev(d_soa, w.layout, d_flags=f["fl"], d_min_dist=f["md"], d_margin=f["mg"],
                                                  d_margin_err=f["me"], d_band=f["bd"]))
    d64, m64, fl64 = md.cpu().numpy(), mg.cpu().numpy(), fl.cpu().numpy()
    d32, m32, e32 = f["md"].cpu().numpy().astype(np.float64), f["mg"].cpu().numpy().astype(np.float64), f["me"].cpu().numpy().astype(np.float64)
    fin = np.isfinite(d64)
    x_max = float(np.abs(d_soa[0:3].cpu().numpy()).max())
    enc = np.isfinite(e32) & np.isfinite(m64)
    diff = (f["fl"].cpu().numpy() != fl64) & ~f["bd"].cpu().numpy().astype(bool)
    fp32 = {"states": n, "ms": f32_ms, "states_per_s": n / f32_ms * 1e3, "dtype": "f32 broadphase + narrowphase and f32 statics with error bounds; f64 chain frames",
            "tolerance": {"min_dist_abs_m": eng.f32_distance_tolerance(x_max),
                          "min_dist_rule": "64 * 2^-24 * (largest |root coordinate| + reach + 1 m), per state (include/mvk.h)",
                          "margin_rule": "per state: the f64 margin lies within margin +- margin_err (certified enclosure)"},
            "observed": {"max_abs_err_min_dist_m": float(np.abs(d32[fin] - d64[fin]).max()) if fin.any() else None,
                         "margin_enclosures_finite": float(enc.mean()), "margin_enclosures_violated": int(np.sum(np.abs(m32[enc] - m64[enc]) > e32[enc] + 4e-7 * np.maximum(np.abs(m64[enc]), 1.0))),
                         "median_margin_err": float(np.median(e32[enc])) if enc.any() else None,
                         "band_fraction": float(f["bd"].float().mean().item()), "flag_mismatches_outside_band": int(diff.sum())}}
    return {"fp32_distances": fp32, "full_path": {"states": n, "ms": full_ms, "states_per_s": n / full_ms * 1e3,
                          "outputs": "flags, min_dist, margin per state + batch statistics",
                          "min_dist": stats["min_dist"], "min_margin": stats["min_margin"], "valid": stats["valid_count"],
                          "roofline": distance_roofline(workload, "distance_kernel", n, full_ms) if workload else None},
            "avoidance_rows": {"states": na, "ms": avoid_ms, "states_per_s": na / avoid_ms * 1e3,
                               "outputs": "per primitive: distance, nearest box and points, constraint row, bound",
                               "roofline": distance_roofline(workload, "avoid_kernel", na, avoid_ms) if workload else None}}


def single_calls(eng, w):
    """host round trips of the calls a one-state-at-a-time caller makes (OMPL's isValid / clearance, one QP iteration
    of the differential-kinematics planner): mean wall time per call over 200 calls on the bench workload's scene"""
    import numpy as np
    s = np.ascontiguousarray(w.soa[:, :1])
    st = np.ascontiguousarray(w.soa[:, 0])

    def per_call(fn, reps=200):
        # warm for 0.2 s first: after an idle spell the GPU serves the first ~50 ms of small launches several times
        # slower (clocks ramping up); a planner calling in a loop never sees that state
        t_w = time.perf_counter()
        while time.perf_counter() - t_w < 0.2:
            fn()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        return (time.perf_counter() - t0) / reps * 1e6
    # one edge of 41 interpolation steps (OMPL's checkMotion): the first state to a neighbour, in the planner's layout
    from motion_planning_amd import _abi as A

    def ompl_row(p):
        if w.params.motion_type == A.SE2:
            return np.concatenate([p[0:2], [2.0 * np.arctan2(p[5], p[6])], p[7:]])
        return p
    e1 = np.ascontiguousarray(ompl_row(st)[:, None])
    e2 = np.ascontiguousarray(ompl_row(0.98 * st + 0.02 * w.soa[:, 1])[:, None])
    one_edge = None
    if e1.shape[0] == eng.state_dim(A.LAYOUT_OMPL):
        one_edge = per_call(lambda: eng.check_motion_batch(e1, e2, 41))
    return {"unit": "us per call (host pointers, ctypes included)",
            "check_motion_one_edge_41_steps": one_edge,
            "is_valid_flag_batch_of_one": per_call(lambda: eng.is_valid_batch(s, w.layout)),
            "is_valid_with_distance_and_margin_batch_of_one": per_call(lambda: eng.is_valid_batch(s, w.layout, want=("flags", "min_dist", "margin", "band"))),
            "avoidance_rows_one_state": per_call(lambda: eng.collision_avoidance_batch(s, w.layout), reps=100)}


def make_workload(name, n):
    import common
    if name == "c2":
        return common.config2(n=n)
    if name == "c2se3":
        return common.config2(n=n, orient="so3")
    if name == "c2k":
        return common.config2k(n=n)
    if name == "c3":
        return common.config3(n=n)
    return common.config5(n=64)   # c5: the sweep generates its states in the kernel, only model / scene / params are used


def roofline_of(workload, w, n, screen_ms, exact_ms, call_ms, sweep):
    """roofline object of the dominant kernel (screen_kernel) of one flag-path launch over n states.

    The path is VALU bound (no dense contraction, ~100 B of HBM traffic per check), so `bound`
    deviates from the hbm|mfma menu on purpose.  frac = executed VALU wave instructions per launch (SQ_INSTS_VALU of
    the committed rocprofv3 PMC pass of this workload, profiles/latest_pmc.json) / the kernel's duration measured
    live / the chip's VALU issue peak: a number <= 1.  `achieved` is the same rate in TFLOP/s, every executed
    instruction counted as an fp64 FMA on 64 lanes (128 flop), against the fp64 vector peak.  The all-pairs model of
    SURVEY.md 8(d) (n * (F_model + P*M*30) flop, most of which the broadphase grid never executes) is reported as
    cull_gain_vs_allpairs; it is not a utilisation."""
    kind = "dragon" if w.model.has_gimbal else "hydrus"
    NL, P, M, J = w.model.n_rotor, w.model.n_prim, len(w.boxes), w.model.n_joint
    f_model = F_MODEL[(kind, NL)]
    f_alg = f_model + P * M * F_PAIR
    b_alg = (0 if sweep else 8 * (7 + J)) + 1.0 / 8.0
    t = screen_ms * 1e-3
    pmc = pmc_traffic(workload) or {}
    stale = pmc.get("stale")
    if stale:
        pmc = {"source": pmc.get("source")}
    insts = pmc.get("valu_insts_per_launch")
    scale = n / pmc["states_per_launch"] if insts and pmc.get("states_per_launch") else 1.0
    rate = insts * scale / t if insts else None
    return {"bound": "valu_fp64", "kernel": "screen_kernel",
            "achieved": rate * 128.0 / 1e12 if rate else None, "peak": PEAK_FP64_VALU_TFLOPS, "unit": "TFLOP/s",
            "frac": rate / PEAK_VALU_WAVE_INSTS_PER_S if rate else None,
            "frac_definition": "executed VALU wave instructions per launch (PMC SQ_INSTS_VALU) / kernel_ms / 6.144e11 per s",
            "valu_wave_insts_per_launch": insts * scale if insts else None,
            "valu_busy_pmc": pmc.get("valu_busy"),
            "traffic": pmc.get("hbm_bytes_per_launch"),
            "stale_pmc": stale,
            "kernel_ms": screen_ms, "second_kernel": "exact_kernel", "second_kernel_ms": exact_ms, "call_ms": call_ms,
            "kernel_ms_note": "kernel_ms / second_kernel_ms come from a separate pass with HIP event markers between the launches "
                              "(mvk_set_profiling), call_ms from events around whole calls; the timed region carries neither, so "
                              "their sum can exceed ms_per_step by a few per cent.  That pass runs one ctx on one stream (launches back to "
                              "back: a launch duration; profiles/<round>/<workload>_summary.txt, first kernel-stats block, is rocprofv3's "
                              "figure for the same) -- in the pipe, where ms_per_step comes from, two contexts' kernels share the chip",
            "bytes_per_check": b_alg, "flop_per_check_uncullable": f_model, "flop_per_check_allpairs": f_alg,
            "achieved_uncullable_tflops": n * f_model / t / 1e12,
            "cull_gain_vs_allpairs": n * f_alg / t / 1e12 / PEAK_FP64_VALU_TFLOPS,
            "hbm": {"achieved": n * b_alg / t / 1e9, "peak": PEAK_HBM_GBS, "unit": "GB/s", "frac": n * b_alg / t / 1e9 / PEAK_HBM_GBS},
            "pmc_source": pmc.get("source")}


def workload_leg(name, n, dev, steps=10, warmup=2):
    """one more single-GPU configuration of BASELINE.json through the same flag path: ms per step over `steps` launches
    (torch events on the launch stream), the per-kernel split, and its own roofline object"""
    import torch

    import common
    import motion_planning_amd as mp
    statics_only = name == "c5_statics"   # the mode-0 sweep of scripts/state_validation_l4.py:46-68 (joints only, no collision)
    if name == "c5_8m":   # BASELINE config 5 at its per-GPU size: 64 M samples over 8 GPUs = 8 M per launch
        n = 1 << 23
    w = make_workload("c5" if name.startswith("c5") else name, n)
    sweep = name.startswith("c5")
    if statics_only:
        from motion_planning_amd import _abi as A
        w.params = mp.default_params(w.model, A.MODE_ONLY_JOINTS, A.SE2)
    eng = mp.Engine(w.model, w.boxes, w.params, device=dev.index)
    d_soa = torch.from_numpy(w.soa).to(dev)
    mask = torch.zeros((n + 63) // 64, dtype=torch.int64, device=dev)
    pose = np.array([0, 0, 0, 0, 0, 0, 1.0])

    def run():
        if sweep:
            eng.sweep_mask_dev(pose, 0, n, common.SEED0 + 5, mask)
        else:
            eng.is_valid_mask_dev(d_soa, w.layout, mask)
    for _ in range(warmup):
        run()
    prewarm(run)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        run()
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    eng.set_profiling(True)
    split = []
    for _ in range(5):
        run()
        split.append(eng.last_kernel_ms())
    screen_ms, exact_ms = float(np.mean([x[0] for x in split])), float(np.mean([x[1] for x in split]))
    bits = np.unpackbits(mask.cpu().numpy().view(np.uint8), bitorder="little")[:n]
    eng.close()
    # the same batches through the library's pipe (two contexts, two streams), as the headline's timed steps
    pipe = mp.Pipe(w.model, w.boxes, w.params, depth=2, device=dev.index)
    pmasks = [torch.zeros_like(mask) for _ in range(2)]

    def prun(k):
        if sweep:
            pipe.sweep_mask(pose, 0, n, common.SEED0 + 5, pmasks[k & 1])
        else:
            pipe.is_valid_mask(d_soa, w.layout, pmasks[k & 1])
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < 0.25:
        for k in range(16):
            prun(k)
        pipe.drain()
    t0 = time.perf_counter()
    psteps = 200   # (wall clock with a drain on both sides: enough steps for the ends not to count)
    for k in range(psteps):
        prun(k)
    pipe.drain()
    pms = (time.perf_counter() - t0) / psteps * 1e3
    psame = bool(torch.equal(pmasks[0], mask) and torch.equal(pmasks[1], mask))
    pipe.close()
    kind = "dragon" if w.model.has_gimbal else "hydrus"
    return {"pipelined": {"entry_point": "mvk_pipe_sweep_mask" if sweep else "mvk_pipe_is_valid_mask", "ms_per_step": pms, "checks_per_s": n / pms * 1e3,
                          "steps": psteps, "masks_equal_one_ctx": psame},
            "workload": f"{name}: {kind.upper()} {w.model.n_rotor}-link, {n} samples, {len(w.boxes)} box obstacles"
                        + (", states generated in-kernel (C-space sweep)" if sweep else "")
                        + (", planning mode 0: joints only, stability verdict alone" if statics_only else ""),
            "ms_per_step": ms, "checks_per_s": n / ms * 1e3, "steps": steps, "valid_fraction": float(bits.mean()),
            "roofline": None if statics_only else roofline_of("c5" if name == "c5_8m" else name, w, n, screen_ms, exact_ms, ms, sweep),
            "kernel_ms": [screen_ms, exact_ms]}


def fp32_variant(eng, w, d_soa, n, dev, steps=10):
    """BASELINE.json north_star: "fp32 variant reported with its own tolerance".  The same batch through
    MVK_PRECISION_F32_REFINED (model update, stability tests and the collision screen's frames in fp32 with error
    bounds; verdicts the bounds cannot guarantee, and every possible contact, are taken again in fp64) next to the
    fp64 flag path: rate, share of states refined, and the number of verdicts that differ."""
    import torch
    from motion_planning_amd import _abi as A
    words = (n + 63) // 64
    m64 = torch.zeros(words, dtype=torch.int64, device=dev)
    m32 = torch.zeros(words, dtype=torch.int64, device=dev)
    band = torch.zeros(n, dtype=torch.uint8, device=dev)
    fl = torch.zeros(n, dtype=torch.uint8, device=dev)
    eng.is_valid_mask_dev(d_soa, w.layout, m64)
    eng.is_valid_batch_dev(d_soa, w.layout, d_flags=fl, d_band=band)     # the full path's boundary band (1e-6)
    eng.set_flag_precision(A.PRECISION_F32_REFINED)
    prewarm(lambda: eng.is_valid_mask_dev(d_soa, w.layout, m32))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(steps):
        eng.is_valid_mask_dev(d_soa, w.layout, m32)
    b.record()
    torch.cuda.synchronize()
    ms = a.elapsed_time(b) / steps
    split = eng.last_kernel_ms()
    refined = eng.last_refined_count()
    eng.set_flag_precision(A.PRECISION_F64)
    b64 = np.unpackbits(m64.cpu().numpy().view(np.uint8), bitorder="little")[:n]
    b32 = np.unpackbits(m32.cpu().numpy().view(np.uint8), bitorder="little")[:n]
    diff = b64 != b32
    inb = band.cpu().numpy().astype(bool)
    return {"dtype": "f32 screen stage, uncertain verdicts and possible contacts refined in f64",
            "ms_per_step": ms, "checks_per_s": n / ms * 1e3, "screen_stage_ms": split[0], "exact_stage_ms": split[1],
            "refined_fraction": refined / n,
            "tolerance": "fp32 stage: slack errors bounded per test by first-order propagation (~1e-4 relative), frame "
                         "positions by <= 1e-3 m; no verdict is taken within those bounds, so flags carry the fp64 path's 1e-6 band only",
            "mismatches_vs_f64_flag_path": int(diff.sum()), "mismatches_outside_band": int((diff & ~inb).sum())}


_FINGERPRINT = None


def kernel_fingerprint():
    global _FINGERPRINT
    if _FINGERPRINT is None:
        sys.path.insert(0, os.path.join(ROOT, "profiles"))
        from fingerprint import kernel_fingerprint as fp
        _FINGERPRINT = fp(ROOT)
    return _FINGERPRINT


def pmc_traffic(workload):
    """Counters of the dominant kernel from the committed rocprofv3 PMC passes (profiles/latest_pmc.json, written by
    profiles/summarize.py; FETCH_SIZE already doubled per the gfx950 correction of MI355X_MICROARCH.md).  An entry is
    used only when it was collected from the kernel sources this run is built from (profiles/fingerprint.py: sha256
    over mvk_kernels.hip, mvk_internal.h, mvk.h, stored at collection time): counters of another kernel version would
    make roofline.frac silently stale.  Returns None (no entry) or {"stale": reason} instead."""
    path = os.path.join(ROOT, "profiles", "latest_pmc.json")
    try:
        d = json.load(open(path)).get(workload)
    except (OSError, ValueError):
        return None
    if d is None:
        return None
    have, want = d.get("kernel_fingerprint"), kernel_fingerprint()
    if have != want:
        return {"stale": f"PMC pass {d.get('source')} was collected from kernel sources {have}, this run is built from {want}: "
                         "re-run profiles/run_profile.sh", "source": d.get("source")}
    return d


def bench_edges(args, world, rank, local_rank, dev):
    """config c4: batched MotionValidator edge checks (64 interpolation steps per edge), HYDRUS 6-link, sharded by
    edge over the ranks (never by step); each step ends with the all-gather of the per-edge verdict bytes."""
    import torch
    import torch.distributed as dist

    import common
    import motion_planning_amd as mp
    E = args.edges
    model, params, boxes, s1, s2, nd = common.config4(n_edges=E)
    if world > 1:   # every rank owns different edges
        rng = common.rng_for(200 + rank)
        s1[3:] = rng.uniform(-1.58, 1.58, s1[3:].shape)
        s2[3:] = np.clip(s1[3:] + rng.normal(scale=0.05, size=s1[3:].shape), -1.58, 1.58)
    eng = mp.Engine(model, boxes, params, device=local_rank)
    d1, d2 = torch.from_numpy(s1).to(dev), torch.from_numpy(s2).to(dev)
    ok = torch.zeros(E, dtype=torch.uint8, device=dev)
    first = torch.zeros(E, dtype=torch.int32, device=dev)
    ok_all = torch.zeros(world * E, dtype=torch.uint8, device=dev) if world > 1 else None

    def step():
        eng.check_motion_batch_dev(d1, d2, nd, ok, first)
        if world > 1:
            dist.all_gather_into_tensor(ok_all, ok)
    for _ in range(args.prewarm + args.warmup):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        emit_line({
            "metric": "state validity checks/sec", "value": world * E * nd * args.steps / elapsed, "unit": "checks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": args.prewarm, "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": f"c4: HYDRUS 6-link edge checks, {E} edges/GPU x {nd} interpolation steps, 100 box obstacles",
                       "edges_per_gpu": E, "steps_per_edge": nd, "parallelism": f"shard{world} (by edge)"},
            "edges_per_s": world * E * args.steps / elapsed, "valid_edge_fraction": float(ok.float().mean().item())})
    eng.close()
    if world > 1:
        dist.destroy_process_group()


_JSON_FD = None


def claim_stdout():
    """stdout carries the one JSON line and nothing else: whatever the libraries print there from now on (RCCL's
    version banner at communicator set-up, for one) goes to stderr"""
    global _JSON_FD
    sys.stdout.flush()
    _JSON_FD = os.dup(1)
    os.dup2(2, 1)


def emit_line(obj):
    data = (json.dumps(obj) + "\n").encode()
    if _JSON_FD is None:
        sys.stdout.write(data.decode())
        sys.stdout.flush()
    else:
        sys.stdout.flush()
        os.write(_JSON_FD, data)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--prewarm", type=int, default=1500,
                    help="untimed steps before the warm-up steps that bring the GPU out of its idle clocks (0: measure from cold)")
    ap.add_argument("--samples", type=int, default=1 << 20)
    ap.add_argument("--workload", default="c2k", choices=["c2", "c2se3", "c2k", "c3", "c4", "c5"],
                    help="headline workload (default: the north star's target, 6-link HYDRUS with 1k obstacles)")
    ap.add_argument("--no-fp32", action="store_true", help="skip the fp32-variant leg")
    ap.add_argument("--no-legs", action="store_true", help="skip the other single-GPU workloads (c2, c3, c5)")
    ap.add_argument("--edges", type=int, default=1 << 16, help="c4: edges per GPU")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-planner", action="store_true")
    ap.add_argument("--no-edges", action="store_true")
    ap.add_argument("--no-distance", action="store_true", help="skip the distance-kernel leg (full path, avoidance rows)")
    ap.add_argument("--exchange", default="mvk", choices=["mvk", "torch"],
                    help="who issues the per-step mask all-gather: the library's C ABI (mvk_gather_mask_dev, RCCL called from "
                         "libmvk.so; default) or torch.distributed (diagnostics)")
    ap.add_argument("--overlap-exchange", action="store_true", help="issue the exchange on a second stream (mvk_exchange_masks_dev)")
    ap.add_argument("--count-every-step", action="store_true", help="count the valid bits of the gathered masks every step, not only after the last")
    ap.add_argument("--one-ctx", action="store_true", help="headline through one ctx on one stream (the round-2 form) instead of mvk_pipe_* with two contexts")
    ap.add_argument("--pipe-depth", type=int, default=2, help="contexts (lanes) of the headline's pipe, 2..4")
    ap.add_argument("--pipe-timeout", type=float, default=90.0, help="N > 1: seconds the pipelined exchange loop may take before the "
                    "line is printed from the one-ctx loop's figures instead (a guard for a path no single-GPU box can rehearse)")
    ap.add_argument("--slots", type=int, default=4, help="exchange buffer sets: how many steps the host may run ahead of the exchange")
    ap.add_argument("--cold-steps", type=int, default=20, help="N = 1: timed steps of the from-cold pass before the pre-warm (0: skip)")
    ap.add_argument("--rehearse-exchange", action="store_true",
                    help="one rank: run the step's RCCL exchange (a group of one) exactly as the N > 1 path issues it")
    args = ap.parse_args()
    claim_stdout()

    import torch
    import torch.distributed as dist

    import common
    import motion_planning_amd as mp

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    exchange = world > 1 or args.rehearse_exchange
    if exchange:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    comm, comm_b = None, None
    depth = max(2, min(4, args.pipe_depth))
    lane_comms = []   # the communicators of lanes 1 .. depth - 1 (lane 0 uses comm)
    if exchange and args.exchange == "mvk":
        # the library's own communicators (csrc/mvk_dist.cpp: ncclCommInitRank from libmvk.so); a 128-byte id travels
        # over the process group the launcher set up, which also serves the barrier and the max-over-ranks below.  Two of
        # them: the pipelined loop issues the all-gather of a batch on the stream of the pipe lane that ran it, and a
        # communicator takes its collectives from one stream at a time -- one communicator per lane.
        import ctypes as C
        from motion_planning_amd import _abi as A
        lib = A.load()

        def make_comm():
            ident = torch.zeros(A.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
            if rank == 0:
                buf = (C.c_ubyte * A.COMM_ID_BYTES)()
                if lib.mvk_comm_unique_id(buf) != 0:
                    raise SystemExit("mvk_comm_unique_id: " + lib.mvk_last_error().decode())
                ident.copy_(torch.frombuffer(bytearray(buf), dtype=torch.uint8))
            dist.broadcast(ident, 0)
            buf = (C.c_ubyte * A.COMM_ID_BYTES)(*ident.cpu().tolist())
            c = C.c_void_p()
            rc = lib.mvk_comm_create(buf, rank, world, local_rank, C.byref(c))
            why = lib.mvk_last_error().decode() if rc != 0 else ""
            # every rank must take the same path: agree on the outcome over the launcher's group
            ok = torch.tensor([1 if rc == 0 else 0], dtype=torch.int32, device=dev)
            dist.all_reduce(ok, op=dist.ReduceOp.MIN)
            if int(ok.item()) == 0:
                if rc == 0:
                    lib.mvk_comm_destroy(c)
                return None, why or "this rank was fine"
            return c, ""
        comm, why = make_comm()
        if comm is None:   # fall back to the torch.distributed exchange (the round-2 form) everywhere
            args.exchange = "torch"
            args.exchange_fallback = f"mvk_comm_create failed on some rank ({why}): torch.distributed exchange instead"
            print(f"rank {rank}: {args.exchange_fallback}", file=sys.stderr)
        elif not args.one_ctx:
            for _ in range(1, depth):   # a communicator per lane of the pipe
                cx, why = make_comm()
                if cx is None:
                    args.exchange_fallback = f"a lane's communicator failed on some rank ({why}): headline through one ctx"
                    print(f"rank {rank}: {args.exchange_fallback}", file=sys.stderr)
                    for c_ in lane_comms:
                        lib.mvk_comm_destroy(c_)
                    lane_comms = []
                    break
                lane_comms.append(cx)
            comm_b = lane_comms[0] if lane_comms else None

    if args.workload == "c4":
        return bench_edges(args, world, rank, local_rank, dev)
    n = args.samples
    w = make_workload(args.workload, n)
    kind = "dragon" if w.model.has_gimbal else "hydrus"
    NL, P, M, J = w.model.n_rotor, w.model.n_prim, len(w.boxes), w.model.n_joint
    if world > 1 and args.workload != "c5":
        # every rank owns a different shard of the batch: reseed the states per rank
        rng = common.rng_for(100 + rank)
        w.soa[7:] = rng.uniform(-1.58, 1.58, (J, n))

    eng = mp.Engine(w.model, w.boxes, w.params, device=local_rank)
    eng.set_profiling(True)   # HIP events on the launch stream around each kernel of a call
    d_soa = torch.from_numpy(w.soa).to(dev)   # (c5: unused, the sweep reads no input)
    words = (n + 63) // 64
    # Four sets of exchange buffers: the mask all-gather and the count of step k run on the exchange stream while the
    # kernels of steps k+1 .. k+3 fill the other sets (collectives overlapped with compute; the host waits for the
    # exchange of step k-4 before it refills a set, which by then has long finished).
    n_slot = max(1, min(8, args.slots)) if exchange else 1
    d_masks = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(n_slot)]
    d_alls = [torch.zeros(world * words, dtype=torch.int64, device=dev) for _ in range(n_slot)] if exchange else None
    d_cnts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(n_slot)]
    d_mask = d_masks[0]

    sweep = args.workload == "c5"   # config 5: states generated in the kernel from the sample index, rank r sweeps [r*n, (r+1)*n)
    sweep_pose = np.array([0, 0, 0, 0, 0, 0, 1.0])

    main_stream = torch.cuda.current_stream()
    side = torch.cuda.Stream(device=dev) if exchange else None   # the exchange's own stream: RCCL call + count kernel

    def run_batch(mask):
        if sweep:
            eng.sweep_mask_dev(sweep_pose, rank * n, n, common.SEED0 + 5, mask, stream=main_stream)
        else:
            eng.is_valid_mask_dev(d_soa, w.layout, mask, stream=main_stream)

    # The host loop must stay well under the kernels' 0.14 ms per step, so the exchange is as few calls as it can
    # be: one all-gather of the packed masks, then the valid count of the whole job taken from the gathered words by
    # every rank itself (no second collective); events are made once, the side stream is current for the whole
    # loop (the kernels are handed the main stream explicitly).
    done = [torch.cuda.Event() for _ in range(n_slot)] if exchange else []
    ready = [torch.cuda.Event() for _ in range(n_slot)] if exchange else []
    used = [False] * n_slot

    if comm is not None:
        side_handle, main_handle = C.c_void_p(side.cuda_stream), C.c_void_p(main_stream.cuda_stream)
        mask_ptrs = [C.c_void_p(t.data_ptr()) for t in d_masks]
        all_ptrs = [C.c_void_p(t.data_ptr()) for t in d_alls]
        cnt_ptrs = [C.c_void_p(t.data_ptr()) for t in d_cnts]

        def gather(slot):
            if lib.mvk_gather_mask_dev(comm, mask_ptrs[slot], words, all_ptrs[slot], side_handle) != 0:
                raise SystemExit("mvk_gather_mask_dev: " + lib.mvk_last_error().decode())
    else:
        def gather(slot):
            dist.all_gather_into_tensor(d_alls[slot], d_masks[slot])   # (current stream = side; the host does not block)

    def step_mvk(k):
        # The step with the exchange issued by the library (csrc/mvk_dist.cpp), nothing of torch in the loop.
        # Default: the all-gather follows the kernels on the launch stream (two ctypes calls per step).  Measured on one
        # GPU with a communicator of one rank (profiles/r03/exchange_rehearsal.txt): 143.9 us per step against 138.0
        # without exchange; handing the exchange to a second stream (--overlap-exchange: event, all-gather, count, event,
        # in mvk_exchange_masks_dev) costs MORE there, 154.8 us -- the event marker in the launch stream and the
        # concurrent kernels outweigh the 6 us they hide -- and the valid count of the whole job, taken from the gathered
        # words by a kernel per step (--count-every-step), another 7 us: it is taken once, after the last step.
        slot = k % n_slot
        if args.overlap_exchange:
            lib.mvk_exchange_wait(comm, slot)
            run_batch(d_masks[slot])
            rc = lib.mvk_exchange_masks_dev(comm, slot, mask_ptrs[slot], words, all_ptrs[slot],
                                            cnt_ptrs[slot] if args.count_every_step else None, main_handle, side_handle)
        else:
            run_batch(d_masks[slot])
            rc = lib.mvk_gather_mask_dev(comm, mask_ptrs[slot], words, all_ptrs[slot], main_handle)
            if rc == 0 and args.count_every_step:
                rc = lib.mvk_mask_count_dev(all_ptrs[slot], world * words * 64, cnt_ptrs[slot], main_handle)
        if rc != 0:
            raise SystemExit("exchange: " + lib.mvk_last_error().decode())

    def step(k, ev=None):
        if comm is not None and ev is None:
            return step_mvk(k)
        slot = k % n_slot
        if exchange and used[slot]:   # this set's last exchange (step k-2) must be through before it is refilled
            # a host wait, not a stream dependency: the exchange of two steps ago has long finished, and a wait in the
            # launch stream is a barrier packet between this step's kernels and the last one's (a bubble of several us)
            done[slot].synchronize()
        if ev:
            ev[0].record(main_stream)
        run_batch(d_masks[slot])
        if ev:
            ev[1].record(main_stream)
        if exchange:
            ready[slot].record(main_stream)
            side.wait_event(ready[slot])
            gather(slot)   # RCCL all-gather on the side stream; the host does not block
            eng.mask_count_dev(d_alls[slot], world * words * 64, d_cnts[slot], stream=side)
            done[slot].record(side)
            used[slot] = True

    def drain():
        if comm is not None:
            lib.mvk_exchange_wait(comm, -1)
        for slot in range(n_slot):
            if exchange and used[slot]:
                main_stream.wait_event(done[slot])

    eng.set_profiling(False)   # the library's event records are markers in the queue: not inside the timed region
    cold = None
    if world == 1 and not exchange and args.cold_steps > 0 and args.prewarm > 0:
        # from cold: the driver's W warm-up steps and then `cold_steps` timed ones straight out of the idle spell that
        # building the workload left (clocks still ramping up) -- what a single batch after a pause sees; the headline
        # below is the steady state of a caller that checks batch after batch (print both)
        torch.cuda.synchronize()
        for k in range(args.warmup):
            step(k)
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for k in range(args.cold_steps):
            step(k)
        torch.cuda.synchronize()
        cold = (time.perf_counter() - tc) / args.cold_steps
    # (--prewarm untimed steps first: the GPU's clocks are up before the W warm-up steps and the K timed ones, see prewarm())
    if exchange:
        torch.cuda.set_stream(side)
    for k in range(args.prewarm + args.warmup):
        step(k)
    drain()
    torch.cuda.synchronize()
    if exchange:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(k)
    t_issued = time.perf_counter() - t0   # host time to issue the K steps (diagnostic: a host-bound loop shows here)
    drain()
    torch.cuda.synchronize()
    if exchange:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    # duration of one call on the launch stream, and its per-kernel split (library events on the same stream): sampled
    # outside the timed region, whose queue holds nothing but the kernels (and the exchange)
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(min(args.steps, 10))]
    for k in range(len(ev)):
        step(k, ev[k])
    drain()
    torch.cuda.synchronize()
    call_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    eng.set_profiling(True)
    split = []
    for _ in range(min(args.steps, 10)):
        run_batch(d_mask)
        split.append(eng.last_kernel_ms())
    screen_ms = float(np.mean([x[0] for x in split]))
    exact_ms = float(np.mean([x[1] for x in split]))
    exchange_ms = None
    if exchange:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the exchange on its own (reported separately): mask all-gather + count kernel over the
        # gathered words, back to back on the exchange stream, every rank
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        dist.barrier()
        e0.record(side)
        for _ in range(10):
            gather(0)
            eng.mask_count_dev(d_alls[0], world * words * 64, d_cnts[0], stream=side)
        e1.record(side)
        torch.cuda.synchronize()
        t = torch.tensor([e0.elapsed_time(e1) / 10], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        exchange_ms = float(t.item())
        torch.cuda.set_stream(main_stream)

    # ---- the headline: the same K steps through the library's pipe (include/mvk.h mvk_pipe_*: two contexts, a stream
    # each, batches alternating, so that one batch's exact stage runs beside the next one's screen stage).  With N > 1 the
    # step's mask all-gather follows the batch on its lane's own stream (mvk_gather_mask_dev with mvk_pipe_stream(lane),
    # one communicator per lane): the other lane's kernels run beside it, no event crosses streams, the host never
    # waits.  Every rank runs this loop (it holds collectives); the figures of the one-ctx loop above are kept beside it.
    pipe_res = None
    watchdog = None
    if exchange and not args.one_ctx and comm is not None and comm_b is not None:
        # Two lanes, a communicator each, collectives on two streams: rehearsed with communicators of one rank only (a
        # one-GPU box cannot hold two ranks).  Should that loop ever fail to come back on a real node, the K steps through
        # one ctx above are measured already: print the line from them, say so, and leave.
        import threading

        def bail():
            if rank == 0:
                emit_line({"metric": "state validity checks/sec", "value": world * n * args.steps / elapsed, "unit": "checks/s",
                           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "prewarm_steps": args.prewarm,
                           "ms_per_step": elapsed / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                           "dtype": "f64", "data": "synthetic",
                           "config": {"workload": f"{w.name}: {kind.upper()} {NL}-link, joints+base, {n} samples/GPU, {M} box obstacles, "
                                                  f"{P} robot primitives, flags as packed mask", "samples_per_gpu": n, "obstacles": M,
                                      "links": NL, "parallelism": f"shard{world}"},
                           "pipeline": None,
                           "pipe_fallback": f"the pipelined exchange loop did not finish within {args.pipe_timeout:.0f} s: value is the "
                                            "same K steps through one ctx with the all-gather behind the kernels on the launch stream",
                           "rccl_ranks": lib.mvk_comm_world(comm), "exchange_ms": exchange_ms,
                           "exchange_by": "libmvk.so (mvk_gather_mask_dev -> ncclAllGather, csrc/mvk_dist.cpp)"})
            sys.stdout.flush()
            os._exit(0)
        watchdog = threading.Timer(args.pipe_timeout, bail)
        watchdog.daemon = True
        watchdog.start()
    if not args.one_ctx and (not exchange or (comm is not None and comm_b is not None)):
        import ctypes as C2
        from motion_planning_amd import _abi as A2
        plib = A2.load()
        pipe = mp.Pipe(w.model, w.boxes, w.params, depth=depth, device=local_rank)
        p_slots = depth * max(1, (n_slot + depth - 1) // depth)   # a multiple of the depth: a buffer set always returns to the same lane
        p_masks = [torch.zeros(words, dtype=torch.int64, device=dev) for _ in range(p_slots)]
        p_alls = [torch.zeros(world * words, dtype=torch.int64, device=dev) for _ in range(p_slots)] if exchange else None
        p_streams = [C2.c_void_p(plib.mvk_pipe_stream(pipe.pipe, lane)) for lane in range(depth)]
        p_comms = [comm] + lane_comms

        def pstep(k):
            # a buffer set always returns to the same lane (p_slots is a multiple of the depth), and a lane runs its work in order: the
            # all-gather of step k has left the set's mask long before step k + p_slots writes it -- no event, no host wait
            slot = k % p_slots
            if sweep:
                lane = pipe.sweep_mask(sweep_pose, rank * n, n, common.SEED0 + 5, p_masks[slot])
            else:
                lane = pipe.is_valid_mask(d_soa, w.layout, p_masks[slot])
            if exchange and plib.mvk_gather_mask_dev(p_comms[lane], C2.c_void_p(p_masks[slot].data_ptr()), words,
                                                     C2.c_void_p(p_alls[slot].data_ptr()), p_streams[lane]) != 0:
                raise SystemExit("mvk_gather_mask_dev: " + plib.mvk_last_error().decode())

        def pdrain():
            pipe.drain()
        for k in range(args.prewarm + args.warmup):
            pstep(k)
        pdrain()
        torch.cuda.synchronize()
        if exchange:
            dist.barrier()
        torch.cuda.synchronize()
        tp0 = time.perf_counter()
        for k in range(args.steps):
            pstep(k)
        tp_issued = time.perf_counter() - tp0
        pdrain()
        torch.cuda.synchronize()
        if exchange:
            dist.barrier()
        torch.cuda.synchronize()
        p_elapsed = time.perf_counter() - tp0
        if exchange:
            t = torch.tensor([p_elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            p_elapsed = float(t.item())
        last = (args.steps - 1) % p_slots
        pipe_res = {"elapsed": p_elapsed, "host_issue_ms_per_step": tp_issued / args.steps * 1e3,
                    "masks_equal_one_ctx": bool(torch.equal(p_masks[last], d_masks[(args.steps - 1) % n_slot])) if not sweep or world == 1 else None,
                    "gathered_equals_one_ctx": bool(torch.equal(p_alls[last], d_alls[(args.steps - 1) % n_slot])) if exchange else None}
        pipe.close()
    if watchdog is not None:
        watchdog.cancel()

    if rank == 0:
        valid = int(sum(bin(int(x) & 0xFFFFFFFFFFFFFFFF).count("1") for x in d_mask.cpu().numpy()[:4096]))
        total = world * n * args.steps
        one_ctx_value, one_ctx_ms = total / elapsed, elapsed / args.steps * 1e3
        if pipe_res:
            elapsed = pipe_res["elapsed"]
        value = total / elapsed
        out = {
            "metric": "state validity checks/sec",
            "value": value,
            "unit": "checks/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "prewarm_steps": args.prewarm,
            "ms_per_step": elapsed / args.steps * 1e3,
            "value_cold": n / cold if cold else None,
            "ms_per_step_cold": cold * 1e3 if cold else None,
            "cold_note": f"{args.cold_steps} steps timed right after {args.warmup} warm-up steps, before the {args.prewarm} pre-warm steps: GPU clocks "
                         "still ramping up out of the idle spell of set-up, through one ctx; `value` is the steady state after the pre-warm" if cold else None,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {"workload": f"{w.name}: {kind.upper()} {NL}-link, joints+base, {n} samples/GPU, {M} box obstacles, "
                                   f"{P} robot primitives, flags as packed mask" + (", states generated in-kernel (C-space sweep)" if sweep else ""),
                       "samples_per_gpu": n, "obstacles": M, "links": NL, "parallelism": f"shard{world}"},
            "pipeline": ({"entry_point": "mvk_pipe_sweep_mask" if sweep else "mvk_pipe_is_valid_mask", "contexts": depth, "streams": depth,
                          "note": "the K timed steps go through the library's pipe: batches go round-robin to its contexts, a stream each, "
                                  "so one batch's exact stage runs beside the next one's screen stage (DESIGN.md section 5); per batch the same "
                                  "kernels and bits as one ctx; value_one_ctx / ms_per_step_one_ctx are the same K steps through one ctx on one stream",
                          "masks_equal_one_ctx": pipe_res["masks_equal_one_ctx"], "gathered_equals_one_ctx": pipe_res["gathered_equals_one_ctx"],
                          "host_issue_ms_per_step": pipe_res["host_issue_ms_per_step"]} if pipe_res else None),
            "value_one_ctx": one_ctx_value,
            "ms_per_step_one_ctx": one_ctx_ms,
            "roofline": roofline_of(args.workload, w, n, screen_ms, exact_ms, call_ms, sweep),
            "rccl_ranks": (lib.mvk_comm_world(comm) if comm is not None else dist.get_world_size()) if exchange else 1,
            "valid_in_first_4096_words": valid,
        }
        if exchange:
            last = (args.steps - 1) % n_slot
            if comm is not None and not args.count_every_step:
                eng.mask_count_dev(d_alls[last], world * words * 64, d_cnts[last], stream=main_stream)
                torch.cuda.synchronize()
            out["valid_total_last_step"] = int(d_cnts[last].item())   # all-reduced over ranks (mvk_mask_count_dev + RCCL sum)
            out["exchange_by"] = "libmvk.so (mvk_gather_mask_dev -> ncclAllGather, csrc/mvk_dist.cpp)" if comm is not None else "torch.distributed"
            if getattr(args, "exchange_fallback", None):
                out["exchange_fallback"] = args.exchange_fallback
            out["exchange"] = ("per step: all-gather of the packed masks (RCCL); " +
                               ("on a second stream, overlapped with the next step's kernels; " if args.overlap_exchange or comm is None else "on the launch stream behind the kernels; ") +
                               ("valid count of the whole job from the gathered words every step" if args.count_every_step or comm is None else "valid count of the whole job from the gathered words after the last step"))
            if pipe_res:
                out["exchange_one_ctx"] = out["exchange"]
                out["exchange"] = ("per step: all-gather of the packed masks (RCCL) behind the batch on its pipe lane's own stream "
                                   "(mvk_gather_mask_dev, one communicator per lane; the other lane's kernels run beside it); valid count of "
                                   "the whole job from the gathered words after the last step")
            out["exchange_ms"] = exchange_ms   # one exchange alone (max over ranks): what the overlap hides
            out["host_issue_ms_per_step"] = t_issued / args.steps * 1e3   # rank 0's host loop: launches + exchange calls
        def leg(name, fn):
            # the legs beside the headline: one that fails says so in the line (and on stderr) instead of taking the
            # headline, which is already measured, down with it
            try:
                out[name] = fn()
            except Exception as e:   # noqa: BLE001
                import traceback
                traceback.print_exc(file=sys.stderr)
                out[name] = {"error": f"{type(e).__name__}: {e}"}

        def single():
            eng.set_profiling(False)   # (the library's event records around each kernel would be part of every call)
            try:
                return single_calls(eng, w)
            finally:
                eng.set_profiling(True)
        if not args.no_cpu_baseline and world == 1:   # rank 0 at N = 1 only
            leg("cpu_baseline", lambda: cpu_baseline(w))
        if not args.no_planner and world == 1:
            leg("rrt_connect", rrt_connect_gap_pass)
        if not args.no_edges and world == 1:
            leg("edge_checks", lambda: edge_checks(dev))
        if not args.no_legs and world == 1:
            out["workloads"] = {}
            for k in ("c2", "c3", "c5", "c5_8m", "c5_statics"):
                if k != args.workload:
                    leg("_wl", lambda k=k: workload_leg(k, n, dev))
                    out["workloads"][k] = out.pop("_wl")
        if not args.no_distance and world == 1 and not sweep:
            leg("distance_path", lambda: distance_path(eng, w, d_soa, dev, args.workload))
            leg("single_calls", single)
        if not args.no_fp32 and world == 1 and not sweep:
            leg("fp32_variant", lambda: fp32_variant(eng, w, d_soa, n, dev))
        emit_line(out)
    eng.close()
    for c_ in lane_comms:
        lib.mvk_comm_destroy(c_)
    if comm is not None:
        lib.mvk_comm_destroy(comm)
    if exchange:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
